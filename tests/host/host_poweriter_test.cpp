// One power iteration's LOCAL segments (gwas/pca.go:339-353), device resident, with REAL keys from a toy secret, at a shape with two block rows and two
// (ragged) block columns, bootstraps included:
//   A  QXtLazyNormStream (matmult.go:83-116):  prod = MatMult4StreamCompute(Q, 5, cache)  ->  BootstrapMatAll (here one party: GenShares -> own shares are
//      the aggregate -> Decrypt/Recode/Recrypt at the TARGET scale, mhe.go:315,330)  ->  out - (Q 1) m^T, CMult with XStdInv
//      [AggregateCMat + CollectiveBootstrapMat, pca.go:347-348: bootstrap again]
//   B  QXLazyNormStream (matmult.go:27-77):    QS = CMult(Q1, XStdInv), prod = MatMult4StreamCompute(QS, 5, cacheT) -> bootstrap -> out - QSm, MaskTrunc:
//      the full block column stays at level l, the ragged one drops to l - 1 with its own scale (per-ciphertext level / scale, basics.go:110-127).
// Nothing leaves HBM between upload and the dumps.  tests/test_host_mirror.py replays every step with the oracle (all words) and decrypts the final cells
// at their own scales against the plaintext linear algebra.  The oracle is linked here ONLY to derive the rotation / relinearisation keys from the secret
// (test infrastructure: a Go host would upload cryptoParams.RotKs / Rlk instead).
// Usage: host_poweriter_test <casedir>
#include "../../sfgwas_amd/host/gwas.hpp"
#include "../../oracle/sfgwas_oracle.h"
#include <fstream>
#include <iostream>
static std::vector<uint64_t> readU64(const std::string &fn) {
    std::ifstream f(fn, std::ios::binary | std::ios::ate); if (!f) throw std::runtime_error("cannot open " + fn);
    size_t n = (size_t)f.tellg() / 8; f.seekg(0); std::vector<uint64_t> v(n); f.read((char *)v.data(), n * 8); return v;
}
static void writeU64(const std::string &fn, const std::vector<uint64_t> &v) { std::ofstream f(fn, std::ios::binary); f.write((const char *)v.data(), v.size() * 8); }
template <class T> static crypto::detail::DevBuf toDev(crypto::CryptoParams *cps, const T *h, size_t n) {
    crypto::detail::DevBuf d(cps, n * sizeof(T));
    cps->check(sfg_memcpy_h2d(cps->ctx, d.p, h, n * sizeof(T)), "h2d");
    return d;
}
struct Rand { std::vector<uint64_t> mask, crs; std::vector<int32_t> e0, e1; };
// bootstrap of a whole matrix with one party: the aggregated shares are the party's own (mhe.go:313-331)
static crypto::DevCipherMatrix bootstrap1(crypto::CryptoParams *cps, const crypto::DevCipherMatrix &cm, const std::string &dir, const std::string &tag, int W) {
    const size_t nct = cm.rows * cm.cols, N = (size_t)cps->N();
    auto mask = readU64(dir + "/" + tag + "_mask.bin"), crs = readU64(dir + "/" + tag + "_crs.bin"), e = readU64(dir + "/" + tag + "_e.bin");
    if (mask.size() != nct * N * W || crs.size() != nct * cps->nq * N || e.size() != nct * N) throw std::runtime_error("bootstrap randomness of " + tag + " has the wrong size");
    const int32_t *e32 = (const int32_t *)e.data();
    auto dm = toDev(cps, mask.data(), mask.size()); auto dc = toDev(cps, crs.data(), crs.size());
    auto d0 = toDev(cps, e32, nct * N), d1 = toDev(cps, e32 + nct * N, nct * N);
    mpc::RefreshRandomness rnd; rnd.mask = dm.u(); rnd.maskLimbs = W; rnd.e0 = (const int32_t *)d0.p; rnd.e1 = (const int32_t *)d1.p; rnd.crs = dc.u();
    mpc::RefreshShares sh = mpc::CollectiveBootstrapGenShares(cps, cm, rnd);
    return mpc::CollectiveBootstrapFinish(cps, cm, sh.h0->u(), sh.h1->u(), dc.u());
}
static void dumpMat(const std::string &fn, const crypto::DevCipherMatrix &m, std::ofstream &meta, const std::string &name) {
    writeU64(fn, gwas::flattenCipherMatrix(crypto::ToHost(m)));
    meta << name << " " << m.rows << " " << m.cols << " " << m.level << " " << m.scale << "\n";
}
int main(int argc, char **argv) {
    try {
        const std::string dir = argv[1];
        std::ifstream cs(dir + "/case.txt"); uint64_t n_ind, m_snp, secretSeed; int s, W; cs >> n_ind >> m_snp >> s >> W >> secretSeed;
        auto mod = readU64(dir + "/moduli.bin"); int nq = (int)mod[0], np = (int)mod[1];
        std::vector<uint64_t> qi(mod.begin() + 2, mod.begin() + 2 + nq), pi(mod.begin() + 2 + nq, mod.begin() + 2 + nq + np);
        const double SC = 17179869184.0;
        auto cps = crypto::NewCryptoParams(0, 14, qi, pi, nullptr, SC);
        const int N = cps->N(), slots = cps->GetSlots(), d = 91;
        // ---- keys from the toy secret (oracle as key generator)
        orc_ring *ring = orc_ring_new(14, nq, np, mod.data() + 2, nullptr);
        std::vector<int8_t> sec(N); orc_gen_secret(ring, secretSeed, sec.data());
        const size_t kw = (size_t)orc_rotkeys_beta(ring) * 2 * (nq + np) * N;
        std::vector<uint64_t> key(kw);
        std::vector<int> steps;
        for (int k = 1; k < d; k++) steps.push_back(k);
        for (int g = 1; g * d < slots; g++) steps.push_back(g * d);
        for (int k = 1; k < slots; k *= 2) if (k >= d || true) steps.push_back(k);
        std::sort(steps.begin(), steps.end()); steps.erase(std::unique(steps.begin(), steps.end()), steps.end());
        for (int k : steps) { const uint64_t g = orc_galois_for_rotation(ring, k); orc_gen_rotkey(ring, sec.data(), g, 5000 + k, key.data()); crypto::LoadRotationKey(cps.get(), g, key, false); }
        orc_gen_rlk(ring, sec.data(), 4999, key.data()); crypto::LoadRelinKey(cps.get(), key, false);
        {   // secret-key shard rows [nq][N], NTT domain
            std::vector<uint64_t> sk((size_t)nq * N);
            for (int j = 0; j < nq; j++) { uint64_t *r = sk.data() + (size_t)j * N; for (int x = 0; x < N; x++) r[x] = sec[x] < 0 ? qi[j] - 1 : (uint64_t)sec[x]; orc_ntt(ring, j, r); }
            cps->check(sfg_ctx_load_secret_key(cps->ctx, sk.data(), 0), "load sk");
        }
        orc_ring_free(ring);
        const int nbr_ind = (int)((n_ind - 1) / slots) + 1, mct_snp = (int)((m_snp - 1) / slots) + 1;
        gwas::GenoFileStream gfs(dir + "/geno.bin", n_ind, m_snp, true);
        gwas::MatMult4StreamPreprocess(cps.get(), &gfs, 5, dir + "/cache_G");
        gwas::MatMult4StreamPreprocess(cps.get(), nullptr, 5, dir + "/cache_GT", dir + "/cache_G");
        const int top = nq - 1;
        crypto::DevCipherMatrix Q = crypto::ToDevice(cps.get(), gwas::unflatten(readU64(dir + "/Q.bin"), s, nbr_ind, top, SC, N));
        crypto::DevCipherVector XMean = crypto::ToDevice(cps.get(), gwas::unflatten(readU64(dir + "/XMean.bin"), 1, mct_snp, top, SC, N)[0]);
        crypto::DevCipherVector XStdInv = crypto::ToDevice(cps.get(), gwas::unflatten(readU64(dir + "/XStdInv.bin"), 1, mct_snp, top, SC, N)[0]);
        std::ofstream meta(dir + "/meta.txt"); meta.precision(17);
        // ---- A: QXtLazyNormStream
        crypto::DevCipherMatrix prodA = gwas::MatMult4StreamComputeDev(cps.get(), Q, 5, dir + "/cache_G", mct_snp);
        dumpMat(dir + "/a_prod.bin", prodA, meta, "a_prod");
        crypto::DevCipherMatrix bootA = bootstrap1(cps.get(), prodA, dir, "bootA", W);
        dumpMat(dir + "/a_boot.bin", bootA, meta, "a_boot");
        crypto::DevCipherMatrix outA = gwas::QXtLazyNormStreamLocal2(cps.get(), bootA, Q, XMean, XStdInv, qi);
        dumpMat(dir + "/a_out.bin", outA, meta, "a_out");
        // ---- pca.go:347-348: AggregateCMat (one party: identity) + CollectiveBootstrapMat
        crypto::DevCipherMatrix Q1 = bootstrap1(cps.get(), outA, dir, "bootM", W);
        dumpMat(dir + "/q1.bin", Q1, meta, "q1");
        // ---- B: QXLazyNormStream
        gwas::QXLazyNormState st;
        crypto::DevCipherMatrix prodB = gwas::QXLazyNormStreamLocal1(cps.get(), Q1, dir + "/cache_GT", nbr_ind, XStdInv, qi, st);
        dumpMat(dir + "/b_prod.bin", prodB, meta, "b_prod");
        crypto::DevCipherMatrix bootB = bootstrap1(cps.get(), prodB, dir, "bootB", W);
        dumpMat(dir + "/b_boot.bin", bootB, meta, "b_boot");
        crypto::DevCipherCells fin = gwas::QXLazyNormStreamLocal2(cps.get(), bootB, st, XMean, (int)n_ind, qi);
        for (size_t i = 0; i < fin.size(); i++) for (size_t j = 0; j < fin[i].size(); j++) {
            const auto &c = fin[i][j];
            writeU64(dir + "/b_out_" + std::to_string(i) + "_" + std::to_string(j) + ".bin", crypto::ToHost(c)[0].data);
            meta << "b_out_" << i << "_" << j << " 1 1 " << c.level << " " << c.scale << "\n";
        }
        std::cout << "OK" << std::endl;
        return 0;
    } catch (const std::exception &e) { std::cerr << "ERROR: " << e.what() << std::endl; return 1; }
}
