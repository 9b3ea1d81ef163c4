// Device-resident compositions of the code AROUND the hot products: QXLazyNormStream and
// QXtLazyNormStream (matmult.go:27-116) split at their network call (BootstrapMatAll, which stays in Go: here the identity), and one
// column step of DCMatMulAAtB (matmult.go:121-156) split at AggregateCVec.  Everything between upload and the final download stays
// in HBM.  tests/test_host_mirror.py builds the inputs and replays the same compositions with the oracle.
// Usage: host_lazynorm_test <casedir>
#include "../../sfgwas_amd/host/gwas.hpp"
#include <fstream>
#include <iostream>
static std::vector<uint64_t> readU64(const std::string &fn) {
    std::ifstream f(fn, std::ios::binary | std::ios::ate); if (!f) throw std::runtime_error("cannot open " + fn);
    size_t n = (size_t)f.tellg() / 8; f.seekg(0); std::vector<uint64_t> v(n); f.read((char *)v.data(), n * 8); return v;
}
static void writeU64(const std::string &fn, const std::vector<uint64_t> &v) { std::ofstream f(fn, std::ios::binary); f.write((const char *)v.data(), v.size() * 8); }
static crypto::CipherVector readVec(const std::string &fn, int n, int level, double scale, int N) {
    auto flat = readU64(fn); return gwas::unflatten(flat, 1, n, level, scale, N)[0];
}
int main(int argc, char **argv) {
    try {
        const std::string dir = argv[1];
        std::ifstream cs(dir + "/case.txt"); uint64_t nrow, ncol; int s, qlevel; cs >> nrow >> ncol >> s >> qlevel;
        auto mod = readU64(dir + "/moduli.bin"); int nq = (int)mod[0], np = (int)mod[1];
        std::vector<uint64_t> qi(mod.begin() + 2, mod.begin() + 2 + nq), pi(mod.begin() + 2 + nq, mod.begin() + 2 + nq + np);
        const double SC = 17179869184.0;
        auto cps = crypto::NewCryptoParams(0, 14, qi, pi, nullptr, SC);
        const int N = cps->N(), beta = (nq + np - 1) / np; const size_t kw = (size_t)beta * 2 * (nq + np) * N;
        auto keys = readU64(dir + "/keys.bin");
        for (size_t k = 0, off = 1; k < keys[0]; k++, off += 1 + kw) crypto::LoadRotationKey(cps.get(), keys[off], std::vector<uint64_t>(keys.begin() + off + 1, keys.begin() + off + 1 + kw), false);
        crypto::LoadRelinKey(cps.get(), readU64(dir + "/rlk.bin"), false);
        const int slots = cps->GetSlots(), nbr = (int)((nrow - 1) / slots) + 1, m_ct = (int)((ncol - 1) / slots) + 1;
        gwas::GenoFileStream gfs(dir + "/geno.bin", nrow, ncol, true);
        gwas::MatMult4StreamPreprocess(cps.get(), &gfs, 5, dir + "/cache_X");
        gwas::MatMult4StreamPreprocess(cps.get(), nullptr, 5, dir + "/cache_XT", dir + "/cache_X");
        // ---- QXLazyNormStream: Q is kp x (rows of X) here (the product's operand orientation), XStdInv / XMean live on the same axis
        auto Qh = gwas::unflatten(readU64(dir + "/Q.bin"), s, nbr, qlevel, SC, N);
        crypto::DevCipherMatrix Q = crypto::ToDevice(cps.get(), Qh);
        crypto::DevCipherVector XStdInv = crypto::ToDevice(cps.get(), readVec(dir + "/XStdInv.bin", nbr, qlevel, SC, N));
        crypto::DevCipherVector XMean = crypto::ToDevice(cps.get(), readVec(dir + "/XMean.bin", nbr, qlevel - 1, SC, N));
        gwas::QXLazyNormState st;
        crypto::DevCipherMatrix prod = gwas::QXLazyNormStreamLocal1(cps.get(), Q, dir + "/cache_X", m_ct, XStdInv, qi, st);
        writeU64(dir + "/qx_part1.bin", gwas::flattenCipherMatrix(crypto::ToHost(prod)));
        // BootstrapMatAll (mhe.go:351) would refresh `prod` here over the network; the test passes it through unchanged
        crypto::DevCipherCells fin = gwas::QXLazyNormStreamLocal2(cps.get(), prod, st, XMean, (int)ncol, qi);
        std::cout << "QX level " << fin[0].back().level << std::endl;
        writeU64(dir + "/qx_final.bin", gwas::flattenCipherMatrix(crypto::ToHost(fin)));
        // ---- QXtLazyNormStream on the transposed cache: Q2 is kp x (cols of X)
        auto Q2h = gwas::unflatten(readU64(dir + "/Q2.bin"), s, m_ct, qlevel, SC, N);
        crypto::DevCipherMatrix Q2 = crypto::ToDevice(cps.get(), Q2h);
        crypto::DevCipherMatrix prod2 = gwas::MatMult4StreamComputeDev(cps.get(), Q2, 5, dir + "/cache_XT", nbr);
        crypto::DevCipherVector XMean2 = crypto::ToDevice(cps.get(), readVec(dir + "/XMean2.bin", nbr, qlevel, SC, N));
        crypto::DevCipherVector XStdInv2 = crypto::ToDevice(cps.get(), readVec(dir + "/XStdInv2.bin", nbr, qlevel, SC, N));
        crypto::DevCipherMatrix fin2 = gwas::QXtLazyNormStreamLocal2(cps.get(), prod2, Q2, XMean2, XStdInv2, qi);
        std::cout << "QXt level " << fin2.level << std::endl;
        writeU64(dir + "/qxt_final.bin", gwas::flattenCipherMatrix(crypto::ToHost(fin2)));
        // ---- one column step of DCMatMulAAtB with innerFn = CMult: A[c] = row 0 of Q, B = the s rows of Q
        crypto::DevCipherVector cTQloc = gwas::DCMatMulAAtBLocal1(cps.get(), Q.row(0), Q, qi);
        writeU64(dir + "/aatb_ctq.bin", gwas::flattenCipherMatrix({crypto::ToHost(cTQloc)}));
        std::vector<crypto::DevCipherVector> out(cTQloc.n);          // AggregateCVec (network) would sum cTQloc over the parties here
        gwas::DCMatMulAAtBLocal2(cps.get(), Q.row(0), cTQloc, out, qi);
        std::vector<uint64_t> flat;
        for (auto &o : out) { auto h = crypto::ToHost(o); for (auto &c : h) flat.insert(flat.end(), c.data.begin(), c.data.end()); }
        writeU64(dir + "/aatb_out.bin", flat);
        std::cout << "OK" << std::endl;
        return 0;
    } catch (const std::exception &e) { std::cerr << "ERROR: " << e.what() << std::endl; return 1; }
}
