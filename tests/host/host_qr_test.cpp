// f-2: the LOCAL segments of one forward and one backward column of NetDQRenc (gwas/qrfact.go:75-216, 236-285) as device-resident sequences on a
// column-encrypted matrix whose columns span TWO ciphertexts (ragged second one); the MPC rounds between the segments are replaced by inputs the test
// supplies (alphaScaled, zNewSqrtInv as fresh ciphertexts; one party, so every aggregate is the local value; BootstrapMatAll = the local halves of the
// collective bootstrap at the target scale).  tests/test_host_mirror.py replays the same sequence with the oracle and compares every word, level and scale.
// Usage: host_qr_test <casedir>
#include "../../sfgwas_amd/host/gwas.hpp"
#include <fstream>
#include <iostream>
static std::vector<uint64_t> readU64(const std::string &fn) {
    std::ifstream f(fn, std::ios::binary | std::ios::ate); if (!f) throw std::runtime_error("cannot open " + fn);
    size_t n = (size_t)f.tellg() / 8; f.seekg(0); std::vector<uint64_t> v(n); f.read((char *)v.data(), n * 8); return v;
}
static void writeU64(const std::string &fn, const std::vector<uint64_t> &v) { std::ofstream f(fn, std::ios::binary); f.write((const char *)v.data(), v.size() * 8); }
template <class T> static crypto::detail::DevBuf toDev(crypto::CryptoParams *cps, const T *h, size_t n) {
    crypto::detail::DevBuf d(cps, n * sizeof(T)); cps->check(sfg_memcpy_h2d(cps->ctx, d.p, h, n * sizeof(T)), "h2d"); return d;
}
static std::ofstream *g_meta;
static void dumpCell(const std::string &dir, const std::string &name, const crypto::DevCipherVector &c) {
    writeU64(dir + "/" + name + ".bin", crypto::ToHost(c)[0].data);
    *g_meta << name << " " << c.level << " " << c.scale << "\n";
}
// one ciphertext through the local halves of the collective bootstrap (its own scale -> Params.Scale(), mhe.go:315,330), randomness index k of `tag`
static crypto::DevCipherVector boot1(crypto::CryptoParams *cps, const crypto::DevCipherVector &c, const std::string &dir, const std::string &tag, int k, int W) {
    const size_t N = (size_t)cps->N();
    auto mask = readU64(dir + "/" + tag + "_mask.bin"), crs = readU64(dir + "/" + tag + "_crs.bin"), e = readU64(dir + "/" + tag + "_e.bin");
    const size_t nct = crs.size() / (cps->nq * N);
    const int32_t *e32 = (const int32_t *)e.data();
    auto dm = toDev(cps, mask.data() + (size_t)k * N * W, N * W); auto dc = toDev(cps, crs.data() + (size_t)k * cps->nq * N, cps->nq * N);
    auto d0 = toDev(cps, e32 + (size_t)k * N, N), d1 = toDev(cps, e32 + (nct + k) * N, N);
    crypto::DevCipherMatrix cm = crypto::NewDevCipherMatrix(cps, 1, 1, c.level, c.scale);
    cps->check(sfg_memcpy_d2d(cps->ctx, cm.buf->u(), c.ptr(), crypto::detail::ctWords(cps, c.level) * 8), "d2d");
    mpc::RefreshRandomness rnd; rnd.mask = dm.u(); rnd.maskLimbs = W; rnd.e0 = (const int32_t *)d0.p; rnd.e1 = (const int32_t *)d1.p; rnd.crs = dc.u();
    mpc::RefreshShares sh = mpc::CollectiveBootstrapGenShares(cps, cm, rnd);
    return mpc::CollectiveBootstrapFinish(cps, cm, sh.h0->u(), sh.h1->u(), dc.u()).row(0);
}
int main(int argc, char **argv) {
    try {
        const std::string dir = argv[1];
        std::ifstream cs(dir + "/case.txt"); int ncols, nct, ctid, slotid, W, level; double totN; cs >> ncols >> nct >> ctid >> slotid >> W >> level >> totN;
        auto mod = readU64(dir + "/moduli.bin"); int nq = (int)mod[0], np = (int)mod[1];
        std::vector<uint64_t> qi(mod.begin() + 2, mod.begin() + 2 + nq), pi(mod.begin() + 2 + nq, mod.begin() + 2 + nq + np);
        const double SC = 17179869184.0;
        auto cps = crypto::NewCryptoParams(0, 14, qi, pi, nullptr, SC);
        const int N = cps->N(), beta = (nq + np - 1) / np; const size_t kw = (size_t)beta * 2 * (nq + np) * N;
        auto keys = readU64(dir + "/keys.bin");
        for (size_t k = 0, off = 1; k < keys[0]; k++, off += 1 + kw) crypto::LoadRotationKey(cps.get(), keys[off], std::vector<uint64_t>(keys.begin() + off + 1, keys.begin() + off + 1 + kw), false);
        crypto::LoadRelinKey(cps.get(), readU64(dir + "/rlk.bin"), false);
        cps->check(sfg_ctx_load_secret_key(cps->ctx, readU64(dir + "/sk.bin").data(), 0), "load sk");
        std::ofstream meta(dir + "/meta.txt"); meta.precision(17); g_meta = &meta;
        auto loadCols = [&](const std::string &fn, int nc) {
            crypto::DevCipherMatrix m = crypto::ToDevice(cps.get(), gwas::unflatten(readU64(fn), nc, nct, level, SC, N));
            std::vector<crypto::CellVec> cols; for (int c = 0; c < nc; c++) cols.push_back(crypto::cellsOf(m.row(c)));
            return cols;
        };
        // ================= forward column
        std::vector<crypto::CellVec> A = loadCols(dir + "/A.bin", ncols);
        crypto::DevCipherVector alphaIn = crypto::ToDevice(cps.get(), gwas::unflatten(readU64(dir + "/alpha.bin"), 1, 1, level, SC, N)[0]);
        crypto::DevCipherVector zinvIn = crypto::ToDevice(cps.get(), gwas::unflatten(readU64(dir + "/zinv.bin"), 1, 1, level, SC, N)[0]);
        dumpCell(dir, "f1_zloc", gwas::NetDQRencF1(cps.get(), A[0]));
        crypto::CellVec uvec = gwas::NetDQRencF2(cps.get(), A[0], alphaIn, zinvIn, true, ctid, slotid);
        for (int ci = 0; ci < nct; ci++) dumpCell(dir, "f2_uvec_" + std::to_string(ci), uvec[ci]);
        std::vector<crypto::DevCipherVector> cTQ = gwas::NetDQRencInner(cps.get(), uvec, A, false, true, ctid, slotid);       // AggregateCVec: one party
        for (int j = 0; j < ncols; j++) dumpCell(dir, "f3_ctq_" + std::to_string(j), cTQ[j]);
        gwas::NetDQRencUpdate(cps.get(), uvec, cTQ, A, std::vector<double>(ncols, -2.0 / totN));
        for (int c = 0; c < ncols; c++) for (int ci = 0; ci < nct; ci++) dumpCell(dir, "f3_A_" + std::to_string(c) + "_" + std::to_string(ci), A[c][ci]);
        // BootstrapMatAll: FlattenLevels (mhe.go:313), then every ciphertext at its own scale
        int lvl = 1 << 30; for (auto &col : A) for (auto &c : col) lvl = std::min(lvl, c.level);
        std::vector<crypto::CellVec> Ab(ncols);
        for (int c = 0; c < ncols; c++) for (int ci = 0; ci < nct; ci++)
            Ab[c].push_back(boot1(cps.get(), A[c][ci].level == lvl ? A[c][ci] : crypto::DropLevelDev(A[c][ci], lvl), dir, "bootF", c * nct + ci, W));
        std::vector<crypto::CellVec> A4 = gwas::NetDQRencF4(cps.get(), Ab, true, ctid, slotid);
        for (int c = 0; c + 1 < ncols; c++) for (int ci = 0; ci < nct; ci++) dumpCell(dir, "f4_A_" + std::to_string(c) + "_" + std::to_string(ci), A4[c][ci]);
        // ================= backward column with the Householder vector just built, on a fresh Q slice
        std::vector<crypto::CellVec> Q = loadCols(dir + "/Q.bin", ncols);
        std::vector<crypto::DevCipherVector> cTQb = gwas::NetDQRencInner(cps.get(), uvec, Q, true, true, ctid, slotid);
        for (int j = 0; j < ncols; j++) dumpCell(dir, "b_ctq_" + std::to_string(j), cTQb[j]);
        std::vector<double> consts(ncols, -2.0 / totN); consts[0] = -2.0 / std::sqrt(totN);
        gwas::NetDQRencUpdate(cps.get(), uvec, cTQb, Q, consts);
        for (int c = 0; c < ncols; c++) for (int ci = 0; ci < nct; ci++) dumpCell(dir, "b_Q_" + std::to_string(c) + "_" + std::to_string(ci), Q[c][ci]);
        std::cout << "OK" << std::endl;
        return 0;
    } catch (const std::exception &e) { std::cerr << "ERROR: " << e.what() << std::endl; return 1; }
}
