// f-1: the local halves of CollectiveBootstrapMat (mpc/mhe.go:289-348) through the host mirror, device resident; C6: FlattenLevels / ConcatCipherMatrix.
// Outputs are compared by tests/test_host_mirror.py with the oracle's restatement of dckks.RefreshProtocol.
// Usage: host_bootstrap_test <casedir>   (cm: rows x 1 ciphertexts at `level`, one of them handed over one level higher)
#include "../../sfgwas_amd/host/gwas.hpp"
#include <fstream>
#include <iostream>
static std::vector<uint64_t> readU64(const std::string &fn) {
    std::ifstream f(fn, std::ios::binary | std::ios::ate); if (!f) throw std::runtime_error("cannot open " + fn);
    size_t n = (size_t)f.tellg() / 8; f.seekg(0); std::vector<uint64_t> v(n); f.read((char *)v.data(), n * 8); return v;
}
template <class T> static crypto::detail::DevBuf toDev(crypto::CryptoParams *cps, const std::vector<T> &h) {
    crypto::detail::DevBuf d(cps, h.size() * sizeof(T));
    cps->check(sfg_memcpy_h2d(cps->ctx, d.p, h.data(), h.size() * sizeof(T)), "h2d");
    return d;
}
static void dumpDev(crypto::CryptoParams *cps, const std::string &fn, const void *dev, size_t words) {
    std::vector<uint64_t> h(words); cps->check(sfg_memcpy_d2h(cps->ctx, h.data(), dev, words * 8), "d2h");
    std::ofstream f(fn, std::ios::binary); f.write((const char *)h.data(), words * 8);
}
int main(int argc, char **argv) {
    try {
        const std::string dir = argv[1];
        std::ifstream cs(dir + "/case.txt"); int rows, level, W; double ctScale; cs >> rows >> level >> W >> ctScale;   // ctScale: the scale the matrix arrives with (a product: A.scale * Delta)
        auto mod = readU64(dir + "/moduli.bin"); int nq = (int)mod[0], np = (int)mod[1];
        std::vector<uint64_t> qi(mod.begin() + 2, mod.begin() + 2 + nq), pi(mod.begin() + 2 + nq, mod.begin() + 2 + nq + np);
        const double SC = 17179869184.0;
        auto cps = crypto::NewCryptoParams(0, 14, qi, pi, nullptr, SC);
        const size_t N = (size_t)cps->N();
        cps->check(sfg_ctx_load_secret_key(cps->ctx, readU64(dir + "/sk.bin").data(), 0), "load sk");
        // cm arrives with its first ciphertext one level higher: FlattenLevels (mhe.go:313) drops it
        auto hi = gwas::unflatten(readU64(dir + "/cm_first_hi.bin"), 1, 1, level + 1, ctScale, (int)N);
        auto lo = gwas::unflatten(readU64(dir + "/cm_rest.bin"), rows - 1, 1, level, ctScale, (int)N);
        crypto::CipherMatrix cm; cm.push_back(hi[0]); for (auto &r : lo) cm.push_back(r);
        auto flat = crypto::FlattenLevels(cps.get(), cm);
        if (flat.second != level) throw std::runtime_error("FlattenLevels: wrong minimum level");
        crypto::DevCipherMatrix dcm = crypto::ToDevice(cps.get(), flat.first);
        dumpDev(cps.get(), dir + "/flat.bin", dcm.buf->u(), (size_t)rows * 2 * (level + 1) * N);
        // ConcatCipherMatrix of the matrix with itself: rows x 2
        auto cc = crypto::ConcatCipherMatrixDev(cps.get(), {dcm, dcm});
        if (cc.cols != 2 || cc.rows != (size_t)rows) throw std::runtime_error("ConcatCipherMatrix: wrong shape");
        dumpDev(cps.get(), dir + "/concat.bin", cc.buf->u(), (size_t)rows * 2 * 2 * (level + 1) * N);
        auto mask = toDev(cps.get(), readU64(dir + "/mask.bin"));
        auto crs = toDev(cps.get(), readU64(dir + "/crs.bin"));
        auto e = readU64(dir + "/e.bin");                      // e0 then e1 as int32 pairs packed in u64 words
        std::vector<int32_t> e32((const int32_t *)e.data(), (const int32_t *)e.data() + e.size() * 2);
        std::vector<int32_t> e0(e32.begin(), e32.begin() + rows * N), e1(e32.begin() + rows * N, e32.begin() + 2 * rows * N);
        auto de0 = toDev(cps.get(), e0), de1 = toDev(cps.get(), e1);
        mpc::RefreshRandomness rnd; rnd.mask = mask.u(); rnd.maskLimbs = W; rnd.e0 = (const int32_t *)de0.p; rnd.e1 = (const int32_t *)de1.p; rnd.crs = crs.u();
        mpc::RefreshShares sh = mpc::CollectiveBootstrapGenShares(cps.get(), dcm, rnd);
        dumpDev(cps.get(), dir + "/h0.bin", sh.h0->u(), (size_t)rows * (level + 1) * N);
        dumpDev(cps.get(), dir + "/h1.bin", sh.h1->u(), (size_t)rows * nq * N);
        // "aggregation": the test supplies the other party's shares already added
        auto h0agg = toDev(cps.get(), readU64(dir + "/h0agg.bin")); auto h1agg = toDev(cps.get(), readU64(dir + "/h1agg.bin"));
        crypto::DevCipherMatrix out = mpc::CollectiveBootstrapFinish(cps.get(), dcm, h0agg.u(), h1agg.u(), crs.u());
        if (out.level != nq - 1) throw std::runtime_error("bootstrap output is not at MaxLevel");
        if (out.scale != SC) throw std::runtime_error("bootstrap output is not at Params.Scale()");
        dumpDev(cps.get(), dir + "/out.bin", out.buf->u(), (size_t)rows * 2 * nq * N);
        // eval.MultByConstAndAdd through the mirror: cases "<constant> <level0> <scale0> <levelOut> <scaleOut>" on 2-ciphertext vectors
        {
            std::ifstream mc(dir + "/mbca_cases.txt"); int ncase; mc >> ncase;
            std::ofstream res(dir + "/mbca_scales.txt"); res.precision(17);
            for (int k = 0; k < ncase; k++) {
                double c, s0, sO; int l0, lO; mc >> c >> l0 >> s0 >> lO >> sO;
                auto a = gwas::unflatten(readU64(dir + "/mbca_in_" + std::to_string(k) + ".bin"), 1, 2, l0, s0, (int)N)[0];
                auto o = gwas::unflatten(readU64(dir + "/mbca_out_" + std::to_string(k) + ".bin"), 1, 2, lO, sO, (int)N)[0];
                crypto::DevCipherVector da = crypto::ToDevice(cps.get(), a), dout = crypto::ToDevice(cps.get(), o);
                crypto::MultByConstAndAddDev(cps.get(), da, c, dout, qi);
                dumpDev(cps.get(), dir + "/mbca_res_" + std::to_string(k) + ".bin", dout.ptr(), (size_t)2 * 2 * (dout.level + 1) * N);
                res << dout.level << " " << dout.scale << "\n";
            }
        }
        std::cout << "OK" << std::endl;
        return 0;
    } catch (const std::exception &e) { std::cerr << "ERROR: " << e.what() << std::endl; return 1; }
}
