// A14: the ciphertext x ciphertext matrix helpers of the logistic-regression path (matmult.go:1915-2066) composed from
// device-resident evaluator ops; outputs are compared by tests/test_host_mirror.py with the same compositions of oracle functions.
// Usage: host_cmat_test <casedir>   (M: 2 x 1 ciphertexts, N: 2 x 1 ciphertexts at level qlevel)
#include "../../sfgwas_amd/host/gwas.hpp"
#include <fstream>
#include <iostream>
static std::vector<uint64_t> readU64(const std::string &fn) {
    std::ifstream f(fn, std::ios::binary | std::ios::ate); if (!f) throw std::runtime_error("cannot open " + fn);
    size_t n = (size_t)f.tellg() / 8; f.seekg(0); std::vector<uint64_t> v(n); f.read((char *)v.data(), n * 8); return v;
}
static void dump(const std::string &fn, const std::vector<crypto::DevCipherVector> &vs) {
    std::vector<uint64_t> flat;
    for (auto &v : vs) { auto h = crypto::ToHost(v); for (auto &c : h) flat.insert(flat.end(), c.data.begin(), c.data.end()); }
    std::ofstream f(fn, std::ios::binary); f.write((const char *)flat.data(), flat.size() * 8);
}
int main(int argc, char **argv) {
    try {
        const std::string dir = argv[1];
        std::ifstream cs(dir + "/case.txt"); int rows, qlevel, mcols; cs >> rows >> qlevel >> mcols;
        auto mod = readU64(dir + "/moduli.bin"); int nq = (int)mod[0], np = (int)mod[1];
        std::vector<uint64_t> qi(mod.begin() + 2, mod.begin() + 2 + nq), pi(mod.begin() + 2 + nq, mod.begin() + 2 + nq + np);
        const double SC = 17179869184.0;
        auto cps = crypto::NewCryptoParams(0, 14, qi, pi, nullptr, SC);
        const int N = cps->N(), beta = (nq + np - 1) / np; const size_t kw = (size_t)beta * 2 * (nq + np) * N;
        auto keys = readU64(dir + "/keys.bin");
        for (size_t k = 0, off = 1; k < keys[0]; k++, off += 1 + kw) crypto::LoadRotationKey(cps.get(), keys[off], std::vector<uint64_t>(keys.begin() + off + 1, keys.begin() + off + 1 + kw), false);
        crypto::LoadRelinKey(cps.get(), readU64(dir + "/rlk.bin"), false);
        crypto::DevCipherMatrix M = crypto::ToDevice(cps.get(), gwas::unflatten(readU64(dir + "/M.bin"), rows, 1, qlevel, SC, N));
        crypto::DevCipherMatrix Nm = crypto::ToDevice(cps.get(), gwas::unflatten(readU64(dir + "/N.bin"), rows, 1, qlevel, SC, N));
        dump(dir + "/innerprod.bin", gwas::CMultMatInnerProdDev(cps.get(), M, Nm, qi));
        dump(dir + "/innerprod_vec.bin", {gwas::CMultMatInnerProdVectorDev(cps.get(), M, Nm.row(0), mcols, qi)});
        dump(dir + "/col_col.bin", gwas::CMultMatColTimesToColDev(cps.get(), M, Nm, rows, false, qi));
        dump(dir + "/col_row.bin", gwas::CMultMatColTimesToColDev(cps.get(), M, Nm, rows, true, qi));
        // crypto.CInverse (basics.go:627-640) on the column of M, 3 iterations
        crypto::DevCipherVector col = crypto::NewDevCipherVector(cps.get(), (size_t)rows, qlevel, SC);
        for (int i = 0; i < rows; i++) cps->check(sfg_memcpy_d2d(cps->ctx, col.ptr(i), M.row(i).ptr(), crypto::detail::ctWords(cps.get(), qlevel) * 8), "d2d");
        dump(dir + "/inverse.bin", {crypto::CInverseDev(cps.get(), col, 3, qi)});
        std::cout << "OK" << std::endl;
        return 0;
    } catch (const std::exception &e) { std::cerr << "ERROR: " << e.what() << std::endl; return 1; }
}
