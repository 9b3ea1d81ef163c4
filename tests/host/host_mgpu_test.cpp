// The host mirror on a party that owns several GPUs: crypto::NewCryptoParamsMulti + the unchanged call sites of pca.go:112-113,344,352 -
// MatMult4StreamPreprocess for X and X^T, MatMult4StreamCompute for Q X and Q' X^T - on files prepared by tests/test_host_mirror.py, which compares the outputs
// with the oracle.  Usage: host_mgpu_test <casedir> <devices, e.g. 0,0,0>.  The evaluator ops between the products run on devices[0]'s context as before.
#include "../../sfgwas_amd/host/gwas.hpp"
#include <fstream>
#include <iostream>
#include <sstream>
static std::vector<uint64_t> readU64(const std::string &fn) {
    std::ifstream f(fn, std::ios::binary | std::ios::ate); if (!f) throw std::runtime_error("cannot open " + fn);
    size_t n = (size_t)f.tellg() / 8; f.seekg(0); std::vector<uint64_t> v(n); f.read((char *)v.data(), n * 8); return v;
}
static void writeU64(const std::string &fn, const std::vector<uint64_t> &v) { std::ofstream f(fn, std::ios::binary); f.write((const char *)v.data(), v.size() * 8); }
int main(int argc, char **argv) {
    try {
        const std::string dir = argv[1];
        std::vector<int> devices; { std::stringstream ss(argv[2]); std::string t; while (std::getline(ss, t, ',')) devices.push_back(atoi(t.c_str())); }
        std::ifstream cs(dir + "/case.txt"); uint64_t nrow, ncol; int s, level, maxLevel, square; cs >> nrow >> ncol >> s >> level >> maxLevel >> square;
        auto mod = readU64(dir + "/moduli.bin"); int nq = (int)mod[0], np = (int)mod[1];
        std::vector<uint64_t> qi(mod.begin() + 2, mod.begin() + 2 + nq), pi(mod.begin() + 2 + nq, mod.begin() + 2 + nq + np);
        auto cps = crypto::NewCryptoParamsMulti(devices, 14, qi, pi, nullptr, 17179869184.0);
        const int N = cps->N(), beta = (nq + np - 1) / np; const size_t kw = (size_t)beta * 2 * (nq + np) * N;
        auto keys = readU64(dir + "/keys.bin");
        for (size_t k = 0, off = 1; k < keys[0]; k++, off += 1 + kw) crypto::LoadRotationKey(cps.get(), keys[off], std::vector<uint64_t>(keys.begin() + off + 1, keys.begin() + off + 1 + kw), false);
        const int slots = cps->GetSlots(), nbr = (int)((nrow - 1) / slots) + 1, m_ct = (int)((ncol - 1) / slots) + 1;
        crypto::CipherMatrix A = gwas::unflatten(readU64(dir + "/A.bin"), s, nbr, level, 17179869184.0, N);
        crypto::CipherMatrix AT = gwas::unflatten(readU64(dir + "/AT.bin"), s, m_ct, level, 17179869184.0, N);
        gwas::GenoFileStream gfs(dir + "/geno.bin", nrow, ncol, true);
        // pca.go:112: X, read one row at a time into a 1 MiB staging buffer (a few rows per chunk: every chunk is scattered to the three ranks' windows)
        gwas::MatMult4StreamPreprocess(cps.get(), &gfs, maxLevel, dir + "/cache_X", "", 1u << 20);
        // pca.go:113: X^T from ITS OWN file, as the reference has it (gwas.go:597 geno_pca_transpose.bin) - recognised on the devices as the transpose of the resident
        // shards (chunks of 300 kB straddle the ranks' column windows), so no second copy is made
        gwas::GenoFileStream gfsT(dir + "/geno_t.bin", ncol, nrow, true);
        gwas::MatMult4StreamPreprocess(cps.get(), &gfsT, maxLevel, dir + "/cache_XT", "", 300u << 10);
        { std::lock_guard<std::mutex> lk(cps->resident->mu); const auto &e = cps->resident->tab.at(dir + "/cache_XT");
          if (e.flags != SFG_TRANSPOSE || e.owner || e.mg != cps->resident->tab.at(dir + "/cache_X").mg) throw std::runtime_error("X^T was not recognised as the transpose of the resident X"); }
        // an unrelated matrix of X^T's shape is NOT taken for it (one entry differs): it becomes a matrix of its own
        gwas::GenoFileStream gfsU(dir + "/geno_u.bin", ncol, nrow, true);
        gwas::MatMult4StreamPreprocess(cps.get(), &gfsU, maxLevel, dir + "/cache_U", "", 300u << 10);
        { std::lock_guard<std::mutex> lk(cps->resident->mu); const auto &e = cps->resident->tab.at(dir + "/cache_U"); if (e.flags || !e.owner) throw std::runtime_error("a different matrix was taken for X^T"); }
        auto o1 = gwas::MatMult4StreamCompute(cps.get(), A, maxLevel, dir + "/cache_X", m_ct);                // pca.go:344 -> matmult.go:42
        auto o2 = gwas::MatMult4StreamCompute(cps.get(), AT, maxLevel, dir + "/cache_XT", nbr);               // pca.go:352 -> matmult.go:91
        if ((int)o1.size() != s || (int)o1[0].size() != m_ct || (int)o2[0].size() != nbr || o1[0][0].Level() != maxLevel - 1) throw std::runtime_error("bad output shape");
        writeU64(dir + "/out_x.bin", gwas::flattenCipherMatrix(o1));
        writeU64(dir + "/out_xt.bin", gwas::flattenCipherMatrix(o2));
        // an evaluator op between the products still works (device 0's context, keys loaded on every device)
        writeU64(dir + "/rot.bin", crypto::RotateRight(cps.get(), A[0][0], 1).data);
        std::cout << "OK " << sfg_mgpu_transport(cps->mg) << " world " << sfg_mgpu_world(cps->mg) << std::endl;
        return 0;
    } catch (const std::exception &e) { std::cerr << "ERROR: " << e.what() << std::endl; return 1; }
}
