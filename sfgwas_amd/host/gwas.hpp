// gwas.hpp — host-side mirror (C++17, header-only) of the reference's Go interface for the hot path, written on
// top of the C-ABI (include/sfgwas_hip.h).  Go is not available in this image, so this is what stands in for the
// cgo shim of INTEGRATION.md: same names, same argument meaning, same error behaviour (the reference panics on this
// path — matmult.go:361, filestream.go:60,334 — here: std::runtime_error).
//
//   crypto::CryptoParams / Ciphertext / CipherVector / CipherMatrix      crypto/crypto.go:32-60
//   crypto::RotateRight, RotateRightWithEvaluator                        crypto/basics.go:201-224
//   crypto::Mult, CMult, CPMult, CAdd, CSub, CRescale, InnerSumAll       crypto/basics.go:226-292, 386-470, 568-590, 707-720
//   crypto::CMultConst(Rescale/Mat), CAddConst, AddConst, CPAdd, AddPlain  crypto/basics.go:183-199, 472-497, 533-551, 592-611
//   crypto::EncodeFloatVector, Mask, MaskTrunc, MaskWithScaling, CMask   crypto/crypto.go:398-420, basics.go:110-172, 673-693
//   gwas::GenoFileStream                                                 gwas/filestream.go:284-494
//   gwas::DiagCacheStream (reader + writer, reference byte format)       gwas/filestream.go:19-282
//   gwas::MatMult4Stream / MatMult4StreamPreprocess / MatMult4StreamCompute   gwas/matmult.go:914,1043,1238
//   mpc::BeaverMultElemVec / BeaverMultMat                               mpc/beavermult.go:108-147
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>
#include "../../include/sfgwas_hip.h"

namespace crypto {

struct Ciphertext {               // ckks.Ciphertext of degree 1: Value()[k].Coeffs[m][:] flattened as [2][level+1][N]
    int level = 0;
    double scale = 0;
    std::vector<uint64_t> data;
    int Level() const { return level; }
    double Scale() const { return scale; }
};
using CipherVector = std::vector<Ciphertext>;
using CipherMatrix = std::vector<CipherVector>;

// HBM-resident genotype matrices keyed by the reference's cacheFilePrefix (see gwas::MatMult4StreamPreprocess); owned by the
// CryptoParams that created them and shared with its forks - no process-global table
struct ResidentGeno { sfg_geno *g = nullptr; unsigned flags = 0; bool owner = false; sfg_mgeno *mg = nullptr; uint64_t nrow = 0, ncol = 0; };   // mg: sharded by SNP block over the GPUs of a multi-GPU CryptoParams
struct ResidentTable { std::mutex mu; std::map<std::string, ResidentGeno> tab; };

struct CryptoParams {             // crypto.go:32-60 (the parts the hot path touches)
    sfg_ctx *ctx = nullptr;
    sfg_mgpu *mg = nullptr;          // set by NewCryptoParamsMulti: the party's node (one context per device, owned by the engine); ctx is then device 0's context
    sfg_mgpu *mg_root = nullptr;     // forks: the engine of the object they were forked from (not owned) - the sharded matrices of the shared table multiply on it
    int logN = 14, nq = 0, np = 0;
    double scale = 0;
    std::vector<uint64_t> qi;                                        // ciphertext moduli q_0..q_{nq-1} (Params.Qi())
    std::shared_ptr<ResidentTable> resident = std::make_shared<ResidentTable>();
    int N() const { return 1 << logN; }
    int GetSlots() const { return N() / 2; }                         // crypto.go:282-284
    ~CryptoParams() {
        if (!ctx) return;
        if (resident.use_count() == 1) for (auto &kv : resident->tab) if (kv.second.owner) { if (kv.second.mg) sfg_mgpu_geno_free(mg, kv.second.mg); else sfg_geno_free(ctx, kv.second.g); }
        if (mg) sfg_mgpu_destroy(mg);          // destroys the per-device contexts, ctx among them
        else sfg_ctx_destroy(ctx);
    }
    void check(int rc, const char *what) const {
        if (rc) throw std::runtime_error(std::string(what) + ": " + sfg_last_error(ctx));
    }
    // One CryptoParams per concurrent caller, the way the reference checks a private evaluator out of its pool
    // (crypto.go:287-316; ckks.NewEvaluator per goroutine, matmult.go:1110,1200,1371): shares tables, keys and resident
    // matrices with this object, owns its HIP queues and scratch.  Destroy forks before the object that loaded the keys.
    std::unique_ptr<CryptoParams> Fork() {
        auto c = std::make_unique<CryptoParams>();
        check(sfg_ctx_fork(ctx, &c->ctx), "sfg_ctx_fork");
        c->logN = logN; c->nq = nq; c->np = np; c->scale = scale; c->qi = qi; c->resident = resident;
        c->mg_root = mg ? mg : mg_root;
        return c;
    }
};

// after CollectiveInit (gwas.go:212): ring moduli Q then P, lattigo's 2N-th roots (or nullptr), params.Scale()
inline std::unique_ptr<CryptoParams> NewCryptoParams(int device, int logN, const std::vector<uint64_t> &qi, const std::vector<uint64_t> &pi,
                                                     const uint64_t *psi, double scale) {
    auto cps = std::make_unique<CryptoParams>();
    std::vector<uint64_t> mod(qi); mod.insert(mod.end(), pi.begin(), pi.end());
    if (sfg_ctx_create(&cps->ctx, device, logN, (int)qi.size(), (int)pi.size(), mod.data(), psi, scale))
        throw std::runtime_error(std::string("sfg_ctx_create: ") + sfg_last_error(nullptr));
    cps->logN = logN; cps->nq = (int)qi.size(); cps->np = (int)pi.size(); cps->scale = scale; cps->qi = qi;
    return cps;
}
// The same for a party that owns several GPUs of its node (the reference is one OS process per party, run_example.sh:1-12): the library's multi-GPU
// engine (sfg_mgpu_create).  MatMult4StreamPreprocess then shards the matrix by SNP block, MatMult4StreamCompute runs on every device; everything else (evaluator
// ops, MatMult4Stream forks) runs on devices[0]'s context.
inline std::unique_ptr<CryptoParams> NewCryptoParamsMulti(const std::vector<int> &devices, int logN, const std::vector<uint64_t> &qi, const std::vector<uint64_t> &pi,
                                                          const uint64_t *psi, double scale) {
    auto cps = std::make_unique<CryptoParams>();
    std::vector<uint64_t> mod(qi); mod.insert(mod.end(), pi.begin(), pi.end());
    if (sfg_mgpu_create(&cps->mg, devices.data(), (int)devices.size(), logN, (int)qi.size(), (int)pi.size(), mod.data(), psi, scale))
        throw std::runtime_error(std::string("sfg_mgpu_create: ") + sfg_mgpu_last_error(nullptr));
    cps->ctx = sfg_mgpu_ctx(cps->mg, 0);
    cps->logN = logN; cps->nq = (int)qi.size(); cps->np = (int)pi.size(); cps->scale = scale; cps->qi = qi;
    return cps;
}
// cryptoParams.RotKs (crypto.go:50): one switching key per Galois element, [beta][2][nq+np][N]
inline void LoadRotationKey(CryptoParams *cps, uint64_t galoisEl, const std::vector<uint64_t> &key, bool montgomeryForm) {
    if (cps->mg) { if (sfg_mgpu_load_rotkey(cps->mg, galoisEl, key.data(), montgomeryForm ? 1 : 0)) throw std::runtime_error(std::string("LoadRotationKey: ") + sfg_mgpu_last_error(cps->mg)); return; }
    cps->check(sfg_ctx_load_rotkey(cps->ctx, galoisEl, key.data(), montgomeryForm ? 1 : 0), "LoadRotationKey");
}

inline int Mod(int a, int m) { int r = a % m; return r < 0 ? r + m : r; }

// crypto/basics.go:201-210 (the evaluator argument of the Go signature has no counterpart: the context owns the keys)
inline Ciphertext RotateRightWithEvaluator(CryptoParams *cps, const Ciphertext &ct, int nrot) {
    const size_t words = ct.data.size();
    void *din = nullptr, *dout = nullptr;
    cps->check(sfg_malloc(cps->ctx, &din, words * 8), "RotateRight"); cps->check(sfg_malloc(cps->ctx, &dout, words * 8), "RotateRight");
    cps->check(sfg_memcpy_h2d(cps->ctx, din, ct.data.data(), words * 8), "RotateRight");
    int rc = sfg_rotate_right_dev(cps->ctx, (const uint64_t *)din, (uint64_t *)dout, 1, ct.level, &nrot);
    Ciphertext out; out.level = ct.level; out.scale = ct.scale; out.data.resize(words);
    if (!rc) rc = sfg_memcpy_d2h(cps->ctx, out.data.data(), dout, words * 8);
    sfg_free(cps->ctx, din); sfg_free(cps->ctx, dout);
    cps->check(rc, "RotateRight");
    return out;
}
inline Ciphertext RotateRight(CryptoParams *cps, const Ciphertext &ct, int nrot) { return RotateRightWithEvaluator(cps, ct, nrot); }  // :212-224

// ---- batched evaluator ops (basics.go:226-292, 386-470, 568-590, 707-720)
struct Plaintext { int level = 0; double scale = 0; std::vector<uint64_t> data; };          // NTT domain, [level+1][N]
using PlainVector = std::vector<Plaintext>;
inline void LoadRelinKey(CryptoParams *cps, const std::vector<uint64_t> &key, bool montgomeryForm) {       // cryptoParams.Rlk, crypto.go:47
    cps->check(sfg_ctx_load_relinkey(cps->ctx, key.data(), montgomeryForm ? 1 : 0), "LoadRelinKey");
}
namespace detail {
struct DevBuf {                                                   // device buffer that frees itself
    CryptoParams *cps; void *p = nullptr;
    DevBuf(CryptoParams *c, size_t bytes) : cps(c) { cps->check(sfg_malloc(c->ctx, &p, bytes ? bytes : 8), "sfg_malloc"); }
    ~DevBuf() { if (p) sfg_free(cps->ctx, p); }
    uint64_t *u() const { return (uint64_t *)p; }
};
inline size_t ctWords(const CryptoParams *cps, int level) { return (size_t)2 * (level + 1) * cps->N(); }
inline void upload(CryptoParams *cps, const CipherVector &X, int level, DevBuf &d, bool broadcast, size_t n) {
    const size_t w = ctWords(cps, level);
    for (size_t i = 0; i < n; i++) {
        const Ciphertext &c = X[broadcast ? 0 : i];
        if (c.level != level || c.data.size() != w) throw std::runtime_error("evaluator op: operand levels differ");   // lattigo panics on mixed levels here too
        cps->check(sfg_memcpy_h2d(cps->ctx, d.u() + i * w, c.data.data(), w * 8), "h2d");
    }
}
inline CipherVector download(CryptoParams *cps, const DevBuf &d, size_t n, int level, double scale) {
    const size_t w = ctWords(cps, level);
    CipherVector out(n);
    for (size_t i = 0; i < n; i++) { out[i].level = level; out[i].scale = scale; out[i].data.resize(w); cps->check(sfg_memcpy_d2h(cps->ctx, out[i].data.data(), d.u() + i * w, w * 8), "d2h"); }
    return out;
}
// eval.Rescale(ct, threshold, ct): divide by the last modulus while scale >= threshold * q_level / 2 (lattigo ckks evaluator)
inline CipherVector rescaleOnDevice(CryptoParams *cps, DevBuf &d, size_t n, int level, double scale, double threshold, const std::vector<uint64_t> &qi) {
    if (level == 0) throw std::runtime_error("cannot Rescale: input Ciphertext already at level 0");
    DevBuf t(cps, n * ctWords(cps, level) * 8);
    DevBuf *cur = &d, *nxt = &t;
    while (level != 0 && scale >= threshold * (double)qi[level] / 2) {
        cps->check(sfg_ct_rescale_dev(cps->ctx, cur->u(), nxt->u(), (int)n, level), "Rescale");
        scale /= (double)qi[level]; level--; std::swap(cur, nxt);
    }
    return download(cps, *cur, n, level, scale);
}
}  // namespace detail

// basics.go:226-234: MulRelinNew without rescale
inline Ciphertext Mult(CryptoParams *cps, const Ciphertext &a, const Ciphertext &b) {
    detail::DevBuf da(cps, a.data.size() * 8), db(cps, a.data.size() * 8), dout(cps, a.data.size() * 8);
    detail::upload(cps, {a}, a.level, da, false, 1); detail::upload(cps, {b}, a.level, db, false, 1);
    cps->check(sfg_ct_mulrelin_dev(cps->ctx, da.u(), db.u(), dout.u(), 1, a.level), "Mult");
    return detail::download(cps, dout, 1, a.level, a.scale * b.scale)[0];
}
// basics.go:386-427: element-wise MulRelin + Rescale; a length-1 operand is broadcast. qi = params.Qi()
inline CipherVector CMult(CryptoParams *cps, const CipherVector &X, const CipherVector &Y, const std::vector<uint64_t> &qi) {
    const size_t n = std::max(X.size(), Y.size()); const int level = X[0].level; const size_t w = detail::ctWords(cps, level);
    detail::DevBuf dx(cps, n * w * 8), dy(cps, n * w * 8), dout(cps, n * w * 8);
    detail::upload(cps, X, level, dx, X.size() == 1, n); detail::upload(cps, Y, level, dy, Y.size() == 1 && X.size() != 1, n);
    cps->check(sfg_ct_mulrelin_dev(cps->ctx, dx.u(), dy.u(), dout.u(), (int)n, level), "CMult");
    return detail::rescaleOnDevice(cps, dout, n, level, X[0].scale * Y[0].scale, cps->scale, qi);
}
// basics.go:429-470: ciphertext x plaintext + Rescale, same broadcast rule
inline CipherVector CPMult(CryptoParams *cps, const CipherVector &X, const PlainVector &Y, const std::vector<uint64_t> &qi) {
    const size_t n = std::max(X.size(), Y.size()); const int level = X[0].level; const size_t w = detail::ctWords(cps, level), pw = w / 2;
    detail::DevBuf dx(cps, n * w * 8), dp(cps, Y.size() * pw * 8), dout(cps, n * w * 8);
    detail::upload(cps, X, level, dx, X.size() == 1, n);
    for (size_t i = 0; i < Y.size(); i++) {
        if (Y[i].level != level) throw std::runtime_error("CPMult: operand levels differ");
        cps->check(sfg_memcpy_h2d(cps->ctx, dp.u() + i * pw, Y[i].data.data(), pw * 8), "h2d");
    }
    cps->check(sfg_ct_mul_plain_dev(cps->ctx, dx.u(), dp.u(), Y.size() == 1 ? 0 : pw, dout.u(), (int)n, level), "CPMult");
    return detail::rescaleOnDevice(cps, dout, n, level, X[0].scale * Y[0].scale, cps->scale, qi);
}
// crypto.go:398-420 EncodeFloatVector: packs f into ceil(len/slots) plaintexts at `level` (the reference uses MaxLevel), default scale
inline PlainVector EncodeFloatVector(CryptoParams *cps, const std::vector<double> &f, int level) {
    const size_t slots = (size_t)cps->GetSlots(), nvec = (f.size() + slots - 1) / slots, pw = (size_t)(level + 1) * cps->N();
    std::vector<double> padded(nvec * slots, 0.0);
    std::copy(f.begin(), f.end(), padded.begin());
    detail::DevBuf d(cps, nvec * pw * 8);
    cps->check(sfg_encode_vectors_dev(cps->ctx, padded.data(), (int)nvec, level, d.u()), "EncodeFloatVector");
    PlainVector out(nvec);
    for (size_t i = 0; i < nvec; i++) { out[i].level = level; out[i].scale = cps->scale; out[i].data.resize(pw); cps->check(sfg_memcpy_d2h(cps->ctx, out[i].data.data(), d.u() + i * pw, pw * 8), "d2h"); }
    return out;
}
// basics.go:129-148: keep ? every slot but `ind` : only slot `ind`, times scalingFactor; MulRelinNew(mask, ct) + Rescale
inline Ciphertext MaskWithScaling(CryptoParams *cps, const Ciphertext &ct, int ind, bool keep, double scalingFactor, const std::vector<uint64_t> &qi) {
    std::vector<double> m(cps->GetSlots(), keep ? scalingFactor : 0.0);
    m[ind] = keep ? 0.0 : scalingFactor;
    return CPMult(cps, {ct}, EncodeFloatVector(cps, m, ct.level), qi)[0];
}
inline Ciphertext Mask(CryptoParams *cps, const Ciphertext &ct, int index, bool keepRest, const std::vector<uint64_t> &qi) {   // :150-172
    return MaskWithScaling(cps, ct, index, keepRest, 1.0, qi);
}
inline Ciphertext MaskTrunc(CryptoParams *cps, const Ciphertext &ct, int N, const std::vector<uint64_t> &qi) {                  // :110-127
    if (N == cps->GetSlots()) return ct;
    std::vector<double> m(cps->GetSlots(), 0.0);
    for (int i = 0; i < N; i++) m[i] = 1.0;
    return CPMult(cps, {ct}, EncodeFloatVector(cps, m, ct.level), qi)[0];
}
inline CipherVector CMask(CryptoParams *cps, const CipherVector &cv, int index, bool keepRest, const std::vector<uint64_t> &qi) {   // :673-693
    if (cv.empty()) return cv;
    std::vector<double> m((size_t)cps->GetSlots() * cv.size(), keepRest ? 1.0 : 0.0);     // one mask over the whole vector: every
    m[index] = keepRest ? 0.0 : 1.0;                                                      // ciphertext is multiplied and rescaled
    return CPMult(cps, cv, EncodeFloatVector(cps, m, cv[0].level), qi);
}
inline CipherVector addSub(CryptoParams *cps, const CipherVector &X, const CipherVector &Y, bool sub) {
    const size_t n = X.size(); const int level = X[0].level; const size_t w = detail::ctWords(cps, level);
    if (Y.size() != n) throw std::runtime_error("CAdd/CSub: vector lengths differ");                   // index out of range panic in Go
    detail::DevBuf dx(cps, n * w * 8), dy(cps, n * w * 8);
    detail::upload(cps, X, level, dx, false, n); detail::upload(cps, Y, level, dy, false, n);
    cps->check((sub ? sfg_ct_sub_dev : sfg_ct_add_dev)(cps->ctx, dx.u(), dy.u(), dx.u(), (int)n, level), "CAdd/CSub");
    return detail::download(cps, dx, n, level, X[0].scale);
}
inline CipherVector CAdd(CryptoParams *cps, const CipherVector &X, const CipherVector &Y) { return addSub(cps, X, Y, false); }   // :568-578
inline CipherVector CSub(CryptoParams *cps, const CipherVector &X, const CipherVector &Y) { return addSub(cps, X, Y, true); }    // :580-590
// basics.go:707-720
inline CipherVector CRescale(CryptoParams *cps, const CipherVector &X, const std::vector<uint64_t> &qi) {
    const size_t n = X.size(); const int level = X[0].level;
    detail::DevBuf dx(cps, n * detail::ctWords(cps, level) * 8);
    detail::upload(cps, X, level, dx, false, n);
    return detail::rescaleOnDevice(cps, dx, n, level, X[0].scale, cps->scale, qi);
}
// ---- constants and plaintext addends (basics.go:183-199, 472-497, 533-551, 592-611)
// lattigo ckks scaleUpExact: round(|n * value|) mod q, sign folded back; float64 product and +0.5 (53-bit big.Float), truncation
inline uint64_t scaleUpExact(double value, double n, uint64_t q) {
    const bool neg = value < 0;
    double x = std::floor((neg ? -n * value : n * value) + 0.5);
    const uint64_t res = (uint64_t)std::fmod(x, (double)q);
    return (neg ? q - res : res) % q;
}
// eval.MultByConst with a float64 constant: a fractional constant is scaled by q_level and the ciphertext scale grows by it
inline CipherVector CMultConst(CryptoParams *cps, const CipherVector &X, double constant, const std::vector<uint64_t> &qi) {   // :480-497
    const size_t n = X.size(); if (!n) return X;
    const int level = X[0].level; const size_t w = detail::ctWords(cps, level);
    double scale = 1.0;
    if (constant != 0 && constant - (double)(long long)constant != 0) scale = (double)qi[level];
    std::vector<uint64_t> sc(level + 1, 0);
    for (int m = 0; m <= level; m++) sc[m] = constant != 0 ? scaleUpExact(constant, scale, qi[m]) : 0;
    detail::DevBuf dx(cps, n * w * 8);
    detail::upload(cps, X, level, dx, false, n);
    cps->check(sfg_ct_mul_scalar_dev(cps->ctx, dx.u(), sc.data(), dx.u(), (int)n, level), "CMultConst");
    return detail::download(cps, dx, n, level, X[0].scale * scale);
}
inline CipherVector CMultConstRescale(CryptoParams *cps, const CipherVector &X, double constant, const std::vector<uint64_t> &qi) {   // :533-551
    return CRescale(cps, CMultConst(cps, X, constant, qi), qi);
}
inline CipherMatrix CMultConstMat(CryptoParams *cps, const CipherMatrix &X, double constant, const std::vector<uint64_t> &qi) {       // :472-478
    CipherMatrix res(X.size());
    for (size_t i = 0; i < X.size(); i++) res[i] = CMultConst(cps, X[i], constant, qi);
    return res;
}
// eval.AddConst: round(constant * ct.Scale()) mod q_m on every NTT coefficient of c0
inline CipherVector CAddConst(CryptoParams *cps, const CipherVector &X, double constant, const std::vector<uint64_t> &qi) {            // :604-611
    const size_t n = X.size(); if (!n) return X;
    const int level = X[0].level; const size_t w = detail::ctWords(cps, level);
    std::vector<uint64_t> sc(level + 1, 0);
    for (int m = 0; m <= level; m++) sc[m] = constant != 0 ? scaleUpExact(constant, X[0].scale, qi[m]) : 0;
    detail::DevBuf dx(cps, n * w * 8);
    detail::upload(cps, X, level, dx, false, n);
    cps->check(sfg_ct_add_scalar_dev(cps->ctx, dx.u(), sc.data(), dx.u(), (int)n, level), "CAddConst");
    return detail::download(cps, dx, n, level, X[0].scale);
}
inline Ciphertext AddConst(CryptoParams *cps, const Ciphertext &ct, double constant, const std::vector<uint64_t> &qi) { return CAddConst(cps, {ct}, constant, qi)[0]; }   // :192-199
// eval.AddNew(ct, plaintext)
inline CipherVector CPAdd(CryptoParams *cps, const CipherVector &X, const PlainVector &Y) {                                               // :592-602
    const size_t n = Y.size(); if (!n) return {};
    const int level = X[0].level; const size_t w = detail::ctWords(cps, level), pw = w / 2;
    detail::DevBuf dx(cps, n * w * 8), dp(cps, n * pw * 8);
    detail::upload(cps, X, level, dx, false, n);
    for (size_t i = 0; i < n; i++) {
        if (Y[i].level < level) throw std::runtime_error("CPAdd: plaintext level below the ciphertext level");
        cps->check(sfg_memcpy_h2d(cps->ctx, dp.u() + i * pw, Y[i].data.data(), pw * 8), "h2d");     // first level+1 rows
    }
    cps->check(sfg_ct_add_plain_dev(cps->ctx, dx.u(), dp.u(), pw, dx.u(), (int)n, level), "CPAdd");
    return detail::download(cps, dx, n, level, X[0].scale);
}
inline Ciphertext AddPlain(CryptoParams *cps, const Ciphertext &ct, const Plaintext &pt) { return CPAdd(cps, {ct}, {pt})[0]; }             // :183-190
// basics.go:278-292
inline Ciphertext InnerSumAll(CryptoParams *cps, const CipherVector &X) {
    const size_t n = X.size(); const int level = X[0].level; const size_t w = detail::ctWords(cps, level);
    detail::DevBuf dx(cps, n * w * 8), dout(cps, w * 8);
    detail::upload(cps, X, level, dx, false, n);
    cps->check(sfg_ct_innersum_dev(cps->ctx, dx.u(), (int)n, level, dout.u()), "InnerSumAll");
    return detail::download(cps, dout, 1, level, X[0].scale)[0];
}
inline Ciphertext InnerProd(CryptoParams *cps, const CipherVector &X, const CipherVector &Y, const std::vector<uint64_t> &qi) {   // :274-276
    return InnerSumAll(cps, CMult(cps, X, Y, qi));
}


// ================================================================ device-resident ciphertext vectors / matrices
// The functions above move every operand over PCIe per call (they mirror the Go signatures one to one).  The types below keep
// ciphertexts in HBM between operations, so that the code AROUND the two hot products of a power iteration - lazy
// standardisation (matmult.go:27-116), A * (A^T B) (matmult.go:121-194), InnerProd / Mask chains of the QR - runs without host
// traffic between ops.  A DevCipherVector is n ciphertexts of one level, contiguous: [n][2][level+1][N].
struct DevCipherVector {
    CryptoParams *cps = nullptr;
    std::shared_ptr<detail::DevBuf> buf; size_t off = 0;      // word offset into buf (views of a matrix row share the buffer)
    size_t n = 0; int level = 0; double scale = 0;
    uint64_t *ptr(size_t i = 0) const { return buf->u() + off + i * detail::ctWords(cps, level); }
    size_t size() const { return n; }
};
inline DevCipherVector NewDevCipherVector(CryptoParams *cps, size_t n, int level, double scale) {
    DevCipherVector v; v.cps = cps; v.n = n; v.level = level; v.scale = scale;
    v.buf = std::make_shared<detail::DevBuf>(cps, n * detail::ctWords(cps, level) * 8);
    return v;
}
inline DevCipherVector ToDevice(CryptoParams *cps, const CipherVector &X) {
    DevCipherVector v = NewDevCipherVector(cps, X.size(), X[0].level, X[0].scale);
    detail::upload(cps, X, X[0].level, *v.buf, false, X.size());
    return v;
}
inline CipherVector ToHost(const DevCipherVector &v) {
    CipherVector out(v.n); const size_t w = detail::ctWords(v.cps, v.level);
    for (size_t i = 0; i < v.n; i++) { out[i].level = v.level; out[i].scale = v.scale; out[i].data.resize(w); v.cps->check(sfg_memcpy_d2h(v.cps->ctx, out[i].data.data(), v.ptr(i), w * 8), "d2h"); }
    return out;
}
// s x ncols ciphertexts, one buffer [s][ncols][2][level+1][N]: exactly the layout the product entry points take and return
struct DevCipherMatrix {
    CryptoParams *cps = nullptr; std::shared_ptr<detail::DevBuf> buf; size_t rows = 0, cols = 0; int level = 0; double scale = 0;
    DevCipherVector row(size_t i) const { DevCipherVector v; v.cps = cps; v.buf = buf; v.off = i * cols * detail::ctWords(cps, level); v.n = cols; v.level = level; v.scale = scale; return v; }
};
inline DevCipherMatrix NewDevCipherMatrix(CryptoParams *cps, size_t rows, size_t cols, int level, double scale) {
    DevCipherMatrix m; m.cps = cps; m.rows = rows; m.cols = cols; m.level = level; m.scale = scale;
    m.buf = std::make_shared<detail::DevBuf>(cps, rows * cols * detail::ctWords(cps, level) * 8);
    return m;
}
inline DevCipherMatrix ToDevice(CryptoParams *cps, const CipherMatrix &A) {
    DevCipherMatrix m = NewDevCipherMatrix(cps, A.size(), A[0].size(), A[0][0].level, A[0][0].scale);
    for (size_t i = 0; i < A.size(); i++) { DevCipherVector r = m.row(i); const size_t w = detail::ctWords(cps, m.level);
        for (size_t j = 0; j < A[i].size(); j++) { if (A[i][j].level != m.level) throw std::runtime_error("ToDevice: mixed levels (FlattenLevels first)"); cps->check(sfg_memcpy_h2d(cps->ctx, r.ptr(j), A[i][j].data.data(), w * 8), "h2d"); } }
    return m;
}
inline CipherMatrix ToHost(const DevCipherMatrix &m) { CipherMatrix out(m.rows); for (size_t i = 0; i < m.rows; i++) out[i] = ToHost(m.row(i)); return out; }
// a resident CipherMatrix whose ciphertexts keep their OWN level and scale (crypto.CipherMatrix is [][]*ckks.Ciphertext): [row][col] one-ciphertext vectors
using DevCipherCells = std::vector<std::vector<DevCipherVector>>;
inline CipherMatrix ToHost(const DevCipherCells &c) {
    CipherMatrix out(c.size());
    for (size_t i = 0; i < c.size(); i++) for (const auto &v : c[i]) { CipherVector h = ToHost(v); out[i].insert(out[i].end(), h.begin(), h.end()); }
    return out;
}

// crypto.DropLevel on device (basics.go:806-824)
inline DevCipherVector DropLevelDev(const DevCipherVector &X, int outLevel) {
    DevCipherVector o = NewDevCipherVector(X.cps, X.n, outLevel, X.scale);
    X.cps->check(sfg_ct_drop_level_dev(X.cps->ctx, X.ptr(), o.ptr(), (int)X.n, X.level, outLevel), "DropLevel");
    return o;
}
// crypto.FlattenLevels (basics.go:514-531): every ciphertext dropped to the minimum level of the matrix; returns that level
inline std::pair<CipherMatrix, int> FlattenLevels(CryptoParams *cps, const CipherMatrix &cm) {
    int minLevel = cm[0][0].level; bool mixed = false;
    for (const auto &row : cm) for (const auto &ct : row) { if (ct.level != minLevel) mixed = true; if (ct.level < minLevel) minLevel = ct.level; }
    if (!mixed) return {cm, minLevel};
    CipherMatrix out(cm.size());
    for (size_t i = 0; i < cm.size(); i++) {
        out[i].resize(cm[i].size());
        for (size_t j = 0; j < cm[i].size(); j++) {
            if (cm[i][j].level == minLevel) { out[i][j] = cm[i][j]; continue; }
            DevCipherVector one = DropLevelDev(ToDevice(cps, CipherVector{cm[i][j]}), minLevel);
            out[i][j] = ToHost(one)[0];
        }
    }
    return {out, minLevel};
}
// crypto.ConcatCipherMatrix (basics.go:773-790): column-wise concatenation of the rows of several matrices (device to device)
inline DevCipherMatrix ConcatCipherMatrixDev(CryptoParams *cps, const std::vector<DevCipherMatrix> &parts) {
    if (parts.empty()) throw std::runtime_error("ConcatCipherMatrix: no matrices");
    size_t cols = 0; for (const auto &p : parts) { if (p.rows != parts[0].rows || p.level != parts[0].level) throw std::runtime_error("ConcatCipherMatrix: shapes differ"); cols += p.cols; }
    DevCipherMatrix out = NewDevCipherMatrix(cps, parts[0].rows, cols, parts[0].level, parts[0].scale);
    const size_t w = (size_t)2 * (out.level + 1) * cps->N();
    for (size_t i = 0; i < out.rows; i++) { size_t c0 = 0;
        for (const auto &p : parts) { cps->check(sfg_memcpy_d2d(cps->ctx, out.row(i).ptr(c0), p.row(i).ptr(), p.cols * w * 8), "d2d"); c0 += p.cols; } }
    return out;
}
namespace detail {
// n copies of one ciphertext (the length-1 broadcast of CMult / CSub), device to device
inline DevCipherVector broadcast(const DevCipherVector &x, size_t n) {
    if (x.n == n) return x;
    if (x.n != 1) throw std::runtime_error("evaluator op: vector lengths differ");
    DevCipherVector o = NewDevCipherVector(x.cps, n, x.level, x.scale); const size_t w = ctWords(x.cps, x.level);
    for (size_t i = 0; i < n; i++) x.cps->check(sfg_memcpy_d2d(x.cps->ctx, o.ptr(i), x.ptr(), w * 8), "d2d");
    return o;
}
// eval.Rescale(ct, params.Scale(), ct) on a device vector: while scale >= threshold * q_level / 2 divide by the last modulus
inline DevCipherVector rescaleDev(DevCipherVector x, double threshold, const std::vector<uint64_t> &qi) {
    if (x.level == 0) throw std::runtime_error("cannot Rescale: input Ciphertext already at level 0");
    while (x.level != 0 && x.scale >= threshold * (double)qi[x.level] / 2) {
        DevCipherVector o = NewDevCipherVector(x.cps, x.n, x.level - 1, x.scale / (double)qi[x.level]);
        x.cps->check(sfg_ct_rescale_dev(x.cps->ctx, x.ptr(), o.ptr(), (int)x.n, x.level), "Rescale");
        x = o;
    }
    return x;
}
}  // namespace detail
// lattigo binary ops work at the smaller of the two levels
inline void alignLevels(DevCipherVector &a, DevCipherVector &b) {
    const int l = std::min(a.level, b.level);
    if (a.level != l) a = DropLevelDev(a, l);
    if (b.level != l) b = DropLevelDev(b, l);
}
inline DevCipherVector CMultDev(CryptoParams *cps, DevCipherVector X, DevCipherVector Y, const std::vector<uint64_t> &qi) {     // basics.go:386-427
    const size_t n = std::max(X.n, Y.n);
    alignLevels(X, Y);
    X = detail::broadcast(X, n); Y = detail::broadcast(Y, n);
    DevCipherVector o = NewDevCipherVector(cps, n, X.level, X.scale * Y.scale);
    cps->check(sfg_ct_mulrelin_dev(cps->ctx, X.ptr(), Y.ptr(), o.ptr(), (int)n, X.level), "CMult");
    return detail::rescaleDev(o, cps->scale, qi);
}
inline DevCipherVector CMultScalarDev(CryptoParams *cps, const DevCipherVector &X, const DevCipherVector &ct, const std::vector<uint64_t> &qi) { return CMultDev(cps, X, ct, qi); }   // :553-566
// eval.Add / Sub (CAdd / CSub :568-590; eval.Sub at matmult.go:56,102).  lattigo's evaluateInPlace first matches the scales: the operand with the SMALLER scale is
// multiplied by floor(scale ratio) when that exceeds 1 (an integer MultByConst) and the result carries the larger scale - restated from lattigo v2.1/v2.2
// ckks/evaluator.go, parity unpinned.  It matters on the reference's own path: CMult at level 9 of PN14QP438 does not rescale (q_9 > 2^35, so
// 2^68 < Delta * q_9 / 2), and QXtLazyNormStream then subtracts a scale-2^68 ciphertext from a freshly bootstrapped scale-2^34 one.
inline DevCipherVector CAddSubDev(CryptoParams *cps, DevCipherVector X, DevCipherVector Y, bool sub) {
    const size_t n = std::max(X.n, Y.n);
    alignLevels(X, Y);
    X = detail::broadcast(X, n); Y = detail::broadcast(Y, n);
    auto mulByInt = [&](const DevCipherVector &v, double k, double newScale) {
        std::vector<uint64_t> sc(v.level + 1);
        for (int m = 0; m <= v.level; m++) sc[m] = scaleUpExact(k, 1.0, cps->qi[m]) % cps->qi[m];
        DevCipherVector o = NewDevCipherVector(cps, v.n, v.level, newScale);
        cps->check(sfg_ct_mul_scalar_dev(cps->ctx, v.ptr(), sc.data(), o.ptr(), (int)v.n, v.level), "MultByConst (scale matching)");
        return o;
    };
    double outScale = X.scale;
    if (X.scale > Y.scale && std::floor(X.scale / Y.scale) > 1) Y = mulByInt(Y, std::floor(X.scale / Y.scale), X.scale);
    else if (Y.scale > X.scale && std::floor(Y.scale / X.scale) > 1) { X = mulByInt(X, std::floor(Y.scale / X.scale), Y.scale); outScale = Y.scale; }
    DevCipherVector o = NewDevCipherVector(cps, n, X.level, outScale);
    cps->check((sub ? sfg_ct_sub_dev : sfg_ct_add_dev)(cps->ctx, X.ptr(), Y.ptr(), o.ptr(), (int)n, X.level), "CAdd/CSub");
    return o;
}
inline DevCipherVector InnerSumAllDev(CryptoParams *cps, const DevCipherVector &X) {                                             // :278-292
    DevCipherVector o = NewDevCipherVector(cps, 1, X.level, X.scale);
    cps->check(sfg_ct_innersum_dev(cps->ctx, X.ptr(), (int)X.n, X.level, o.ptr()), "InnerSumAll");
    return o;
}
inline DevCipherVector InnerProdDev(CryptoParams *cps, const DevCipherVector &X, const DevCipherVector &Y, const std::vector<uint64_t> &qi) { return InnerSumAllDev(cps, CMultDev(cps, X, Y, qi)); }   // :274-276
// MaskTrunc on one resident ciphertext (basics.go:110-127): the 0/1 mask is encoded on the device at the ciphertext's level
inline DevCipherVector MaskTruncDev(CryptoParams *cps, const DevCipherVector &ct, int Nkeep, const std::vector<uint64_t> &qi) {
    if (Nkeep == cps->GetSlots()) return ct;
    std::vector<double> m(cps->GetSlots(), 0.0);
    for (int i = 0; i < Nkeep; i++) m[i] = 1.0;
    detail::DevBuf pt(cps, (size_t)(ct.level + 1) * cps->N() * 8);
    cps->check(sfg_encode_vectors_dev(cps->ctx, m.data(), 1, ct.level, pt.u()), "MaskTrunc");
    DevCipherVector o = NewDevCipherVector(cps, ct.n, ct.level, ct.scale * cps->scale);
    cps->check(sfg_ct_mul_plain_dev(cps->ctx, ct.ptr(), pt.u(), 0, o.ptr(), (int)ct.n, ct.level), "MaskTrunc");
    return detail::rescaleDev(o, cps->scale, qi);
}
// ct x plaintext mask built from real slot values (one vector of `slots` doubles per ciphertext, or a single one for all):
// MulRelinNew(mask, ct) + Rescale - the shared tail of Mask / CMask / CPMult (basics.go:110-172, 429-470, 673-693)
inline DevCipherVector mulByRealVectorsDev(CryptoParams *cps, const DevCipherVector &X, const std::vector<double> &vals, const std::vector<uint64_t> &qi) {
    const size_t slots = (size_t)cps->GetSlots(), nvec = vals.size() / slots, pw = (size_t)(X.level + 1) * cps->N();
    if (nvec != 1 && nvec != X.n) throw std::runtime_error("CPMult: plaintext / ciphertext vector lengths differ");
    detail::DevBuf pt(cps, nvec * pw * 8);
    cps->check(sfg_encode_vectors_dev(cps->ctx, vals.data(), (int)nvec, X.level, pt.u()), "EncodeFloatVector");
    DevCipherVector o = NewDevCipherVector(cps, X.n, X.level, X.scale * cps->scale);
    cps->check(sfg_ct_mul_plain_dev(cps->ctx, X.ptr(), pt.u(), nvec == 1 ? 0 : pw, o.ptr(), (int)X.n, X.level), "CPMult");
    return detail::rescaleDev(o, cps->scale, qi);
}
inline DevCipherVector MaskDev(CryptoParams *cps, const DevCipherVector &ct, int index, bool keepRest, const std::vector<uint64_t> &qi) {       // basics.go:150-172 on one ct
    std::vector<double> m(cps->GetSlots(), keepRest ? 1.0 : 0.0); m[index] = keepRest ? 0.0 : 1.0;
    return mulByRealVectorsDev(cps, ct, m, qi);
}
inline DevCipherVector CMaskDev(CryptoParams *cps, const DevCipherVector &cv, int index, bool keepRest, const std::vector<uint64_t> &qi) {      // basics.go:673-693
    std::vector<double> m((size_t)cps->GetSlots() * cv.n, keepRest ? 1.0 : 0.0); m[index] = keepRest ? 0.0 : 1.0;
    return mulByRealVectorsDev(cps, cv, m, qi);
}
// eval.MultByConstAndAdd(ct0, constant, ctOut) on device vectors (pca.go:264: XMean * -meanWeight onto Q[b]; qrfact.go:195,280: vvTA * -2/N onto A / Q):
//   ctOut += ct0 * constant, after lattigo has matched the two scales.  PARITY UNPINNED - the rule is restated from the published lattigo v2.1/v2.2
//   evaluator: a constant with a fractional part is scaled by q_level (as in MultByConst), integers are not; if the receiver's scale is the smaller one it is
//   first multiplied by floor(scale ratio) (an integer MultByConst) and relabelled, if it is the larger one the constant absorbs the ratio.
inline void MultByConstAndAddDev(CryptoParams *cps, const DevCipherVector &ct0, double constant, DevCipherVector &ctOut, const std::vector<uint64_t> &qi) {
    if (ct0.n != ctOut.n) throw std::runtime_error("MultByConstAndAdd: vector lengths differ");
    const int level = std::min(ct0.level, ctOut.level);
    DevCipherVector in = ct0.level > level ? DropLevelDev(ct0, level) : ct0;
    if (ctOut.level > level) ctOut = DropLevelDev(ctOut, level);                     // "forces a drop of ctOut level to ct0 level"
    double scale = 1.0;
    if (constant != 0 && constant - (double)(long long)constant != 0) scale = (double)qi[level];
    auto mulOutByInt = [&](double k) {
        std::vector<uint64_t> sc(level + 1); for (int m = 0; m <= level; m++) sc[m] = scaleUpExact(k, 1.0, qi[m]) % qi[m];
        cps->check(sfg_ct_mul_scalar_dev(cps->ctx, ctOut.ptr(), sc.data(), ctOut.ptr(), (int)ctOut.n, level), "MultByConst(ctOut)");
    };
    if (scale != 1.0) {
        if (ctOut.scale < in.scale * scale) { const double k = std::floor(scale * in.scale / ctOut.scale); if (k > 1) mulOutByInt(k); ctOut.scale = scale * in.scale; }
        else if (ctOut.scale > in.scale * scale) scale = ctOut.scale / in.scale;
    } else {
        if (ctOut.scale > in.scale) scale = ctOut.scale / in.scale;
        else if (in.scale > ctOut.scale) { const double k = std::floor(in.scale / ctOut.scale); if (k > 1) mulOutByInt(k); ctOut.scale = in.scale; }
    }
    std::vector<uint64_t> sc(level + 1, 0);
    for (int m = 0; m <= level; m++) sc[m] = constant != 0 ? scaleUpExact(constant, scale, qi[m]) % qi[m] : 0;
    cps->check(sfg_ct_mul_scalar_add_dev(cps->ctx, in.ptr(), sc.data(), ctOut.ptr(), (int)ctOut.n, level), "MultByConstAndAdd");
}
// eval.AddConstNew / eval.NegNew on device vectors (the receiver keeps its level and scale)
inline DevCipherVector AddConstDev(CryptoParams *cps, const DevCipherVector &X, double constant, const std::vector<uint64_t> &qi) {
    std::vector<uint64_t> sc(X.level + 1, 0);
    for (int m = 0; m <= X.level; m++) sc[m] = constant != 0 ? scaleUpExact(constant, X.scale, qi[m]) : 0;
    DevCipherVector o = NewDevCipherVector(cps, X.n, X.level, X.scale);
    cps->check(sfg_ct_add_scalar_dev(cps->ctx, X.ptr(), sc.data(), o.ptr(), (int)X.n, X.level), "AddConst");
    return o;
}
inline DevCipherVector NegDev(CryptoParams *cps, const DevCipherVector &X, const std::vector<uint64_t> &qi) {
    std::vector<uint64_t> sc(X.level + 1);
    for (int m = 0; m <= X.level; m++) sc[m] = qi[m] - 1;
    DevCipherVector o = NewDevCipherVector(cps, X.n, X.level, X.scale);
    cps->check(sfg_ct_mul_scalar_dev(cps->ctx, X.ptr(), sc.data(), o.ptr(), (int)X.n, X.level), "Neg");
    return o;
}
// crypto.CInverse (basics.go:627-640) = eval.InverseNew(ct, intv.Iter) per ciphertext.  PARITY UNPINNED - lattigo v2.1.0 ckks/algorithms.go restated
// (the fork's source is absent): Goldschmidt iteration for values in (0, 2): cbar = 1 - x, res = 1 + cbar, then steps - 1 times
// cbar = Rescale(cbar^2), res = Rescale((1 + cbar) * res); every product is MulRelin + Rescale(params.Scale()) (= CMultDev), binary ops at the lower level.
inline DevCipherVector CInverseDev(CryptoParams *cps, const DevCipherVector &X, int steps, const std::vector<uint64_t> &qi) {
    DevCipherVector cbar = AddConstDev(cps, NegDev(cps, X, qi), 1.0, qi);
    DevCipherVector res = AddConstDev(cps, cbar, 1.0, qi);
    for (int i = 1; i < steps; i++) {
        cbar = CMultDev(cps, cbar, cbar, qi);
        res = CMultDev(cps, AddConstDev(cps, cbar, 1.0, qi), res, qi);
    }
    return res;
}
inline DevCipherVector viewOne(const DevCipherVector &v, size_t j) { DevCipherVector o = v; o.off = v.off + j * detail::ctWords(v.cps, v.level); o.n = 1; return o; }
// eval.MultByConstNew (crypto.CMultConst, basics.go:480-497): a constant with a fractional part is scaled by q_level and the scale grows by it; no rescale
inline DevCipherVector CMultConstDev(CryptoParams *cps, const DevCipherVector &X, double constant) {
    double scale = 1.0;
    if (constant != 0 && constant - (double)(long long)constant != 0) scale = (double)cps->qi[X.level];
    std::vector<uint64_t> sc(X.level + 1, 0);
    for (int m = 0; m <= X.level; m++) sc[m] = constant != 0 ? scaleUpExact(constant, scale, cps->qi[m]) : 0;
    DevCipherVector o = NewDevCipherVector(cps, X.n, X.level, X.scale * scale);
    cps->check(sfg_ct_mul_scalar_dev(cps->ctx, X.ptr(), sc.data(), o.ptr(), (int)X.n, X.level), "CMultConst");
    return o;
}
// crypto.Rebalance (basics.go:248-255): InnerSumAll of the one ciphertext, times 1 / slots
inline DevCipherVector RebalanceDev(CryptoParams *cps, const DevCipherVector &ct) { return CMultConstDev(cps, InnerSumAllDev(cps, ct), 1.0 / (double)cps->GetSlots()); }

// ---- ckks.Approximate + eval.EvaluateCheby (crypto/basics.go:613-626, 723-768; mpc/mhe.go:634-667 SigmoidApprox, called from gwas/assoc.go:1045) on device vectors.
// PARITY UNPINNED.  The algorithm lives in the lattigo fork, which is absent; this restates the published lattigo v2.1 / v2.2 ckks/polynomial_evaluation.go:
//   * the power basis T_2 .. T_{2^logSplit - 1}, T_{2^logSplit}, .., T_{2^(logDegree-1)} by T_n = 2 T_a T_b - T_{a-b}, a = ceil(n/2), b = floor(n/2): MulRelinNew,
//     Rescale(params.Scale()), Add to itself, AddConst(-1) or Sub;
//   * the recursive split p = q T_{2^k} + r in the Chebyshev basis (q_0 = c_k, q_j = 2 c_{k+j}, r_{k-j} -= c_{k+j}), the scale the q branch must come out with
//     = targetScale * q_level / T.scale so that the Rescale after MulRelin(q, T) lands on targetScale;
//   * the leaves: a zero ciphertext at the level of T_deg and scale targetScale * q_level, AddConst(c_0), MultByGaussianIntegerAndAdd(T_k, int64(c_k * targetScale *
//     q_level / T_k.scale)) for every |c_k| > 1e-14, Rescale;
//   * the re-split of a leading leaf (lead && logSplit > 1 && maxDeg mod 2^(logSplit+1) > 2^(logSplit-1)) with logDegree = bitlen(deg), logSplit = logDegree / 2.
// One thing is NOT lattigo's text: lattigo predicts the level at which MulRelin(q, T) will run from a heuristic on maxDeg that could not be recalled with
// confidence; here that level is computed exactly by a dry run of the same recursion (mulLevel below), which is what the heuristic exists to predict.
struct ChebyPoly { std::vector<double> c; uint64_t maxDeg = 0; bool lead = false; uint64_t Degree() const { return (uint64_t)c.size() - 1; } };
struct ChebyshevInterpolation { ChebyPoly poly; double a = 0, b = 0; };
// ckks.Approximate: interpolation at the degree+1 Chebyshev nodes of [a, b] (real functions only: the reference's Sigmoid, Sqrt, invSqrt, inv).
// libm's cos stands in for Go's math.Cos: the last bit of a node may differ.
template <class F> inline ChebyshevInterpolation Approximate(F f, double a, double b, int degree) {
    const int n = degree + 1;
    std::vector<double> nodes(n), fi(n);
    for (int k = 1; k <= n; k++) nodes[k - 1] = 0.5 * (a + b) + 0.5 * (b - a) * std::cos(((double)k - 0.5) * (3.141592653589793 / (double)n));
    for (int i = 0; i < n; i++) fi[i] = f(nodes[i]);
    ChebyshevInterpolation ch; ch.a = a; ch.b = b; ch.poly.c.assign(n, 0.0); ch.poly.maxDeg = (uint64_t)degree; ch.poly.lead = true;
    for (int i = 0; i < n; i++) {
        const double u = (2 * nodes[i] - a - b) / (b - a);
        double Tprev = 1, T = u;
        for (int j = 0; j < n; j++) { ch.poly.c[j] += fi[i] * Tprev; const double Tnext = 2 * u * T - Tprev; Tprev = T; T = Tnext; }
    }
    ch.poly.c[0] /= (double)n;
    for (int i = 1; i < n; i++) ch.poly.c[i] *= 2.0 / (double)n;
    return ch;
}
inline double Sigmoid(double x) { return 1.0 / (1 + std::exp(-x)); }                  // mhe.go:675-677
namespace detail {
inline uint64_t bitLen(uint64_t x) { uint64_t n = 0; while (x) { n++; x >>= 1; } return n; }
inline void splitCoeffsCheby(const ChebyPoly &p, uint64_t split, ChebyPoly &q, ChebyPoly &r) {
    r = ChebyPoly(); r.c.assign(p.c.begin(), p.c.begin() + split);
    r.maxDeg = p.maxDeg == p.Degree() ? split - 1 : p.maxDeg - (p.Degree() - split + 1);
    q = ChebyPoly(); q.c.assign(p.Degree() - split + 1, 0.0); q.maxDeg = p.maxDeg; q.lead = p.lead;
    q.c[0] = p.c[split];
    for (uint64_t i = split + 1, j = 1; i < p.Degree() + 1; i++, j++) { q.c[i - split] = 2 * p.c[i]; r.c[split - j] -= p.c[i]; }
}
inline DevCipherVector mulRelinDev(CryptoParams *cps, DevCipherVector X, DevCipherVector Y) {           // eval.MulRelin without the Rescale
    alignLevels(X, Y);
    DevCipherVector o = NewDevCipherVector(cps, X.n, X.level, X.scale * Y.scale);
    cps->check(sfg_ct_mulrelin_dev(cps->ctx, X.ptr(), Y.ptr(), o.ptr(), (int)X.n, X.level), "MulRelin");
    return o;
}
struct ChebyEval {
    CryptoParams *cps; std::map<uint64_t, DevCipherVector> C;
    const std::vector<uint64_t> &qi() const { return cps->qi; }
    void powerBasis(uint64_t n) {
        if (C.count(n)) return;
        const uint64_t a = (n + 1) / 2, b = n >> 1, c = a - b;
        powerBasis(a); powerBasis(b); if (c) powerBasis(c);
        DevCipherVector t = CMultDev(cps, C.at(a), C.at(b), qi());                       // MulRelinNew + Rescale(params.Scale())
        t = CAddSubDev(cps, t, t, false);
        t = c == 0 ? AddConstDev(cps, t, -1.0, qi()) : CAddSubDev(cps, t, C.at(c), true);
        C[n] = t;
    }
    static bool resplit(const ChebyPoly &p, uint64_t logSplit) { return p.lead && logSplit > 1 && p.maxDeg % (1ULL << (logSplit + 1)) > (1ULL << (logSplit - 1)); }
    static uint64_t nextPower(const ChebyPoly &p, uint64_t logSplit) { uint64_t np = 1ULL << logSplit; while (np < (p.Degree() >> 1) + 1) np <<= 1; return np; }
    // level the result of recurse() will have (every Rescale of the recursion divides by exactly one modulus: checked where it happens)
    int outLevel(const ChebyPoly &p, uint64_t logSplit, uint64_t logDegree, int *mulLevelOut = nullptr) const {
        if (p.Degree() < (1ULL << logSplit)) {
            if (resplit(p, logSplit)) { const uint64_t ld = bitLen(p.Degree()); return outLevel(p, ld >> 1, ld); }
            return C.at(p.Degree() ? p.Degree() : 1).level - 1;
        }
        const uint64_t np = nextPower(p, logSplit);
        ChebyPoly q, r; splitCoeffsCheby(p, np, q, r);
        int lq = outLevel(q, logSplit, logDegree); const int lr = outLevel(r, logSplit, logDegree);
        if (lq > lr) lq = lr + 1;
        const int lm = std::min(lq, C.at(np).level);
        if (mulLevelOut) *mulLevelOut = lm;
        return lm > lr ? std::min(lm - 1, lr) : std::min(lm, lr) - 1;
    }
    DevCipherVector leaf(double targetScale, const ChebyPoly &p) {
        const DevCipherVector &top = C.at(p.Degree() ? p.Degree() : 1);
        const int level = top.level; const double currentQi = (double)qi()[level];
        DevCipherVector res = NewDevCipherVector(cps, top.n, level, targetScale * currentQi);
        cps->check(sfg_ct_sub_dev(cps->ctx, top.ptr(), top.ptr(), res.ptr(), (int)top.n, level), "zero ciphertext");
        if (std::fabs(p.c[0]) > 1e-14) res = AddConstDev(cps, res, p.c[0], qi());
        for (uint64_t key = p.Degree(); key > 0; key--) {
            if (!(std::fabs(p.c[key]) > 1e-14)) continue;
            const DevCipherVector &T = C.at(key);
            const double constScale = targetScale * currentQi / T.scale;
            const long long cReal = (long long)(p.c[key] * constScale);                  // Go's int64(): toward zero
            std::vector<uint64_t> sc(level + 1);
            for (int m = 0; m <= level; m++) { const long long q = (long long)qi()[m]; sc[m] = (uint64_t)(((cReal % q) + q) % q); }
            DevCipherVector Tl = T.level > level ? DropLevelDev(T, level) : T;
            cps->check(sfg_ct_mul_scalar_add_dev(cps->ctx, Tl.ptr(), sc.data(), res.ptr(), (int)res.n, level), "MultByGaussianIntegerAndAdd");
        }
        return rescaleOnce(res);
    }
    DevCipherVector rescaleOnce(const DevCipherVector &x) {
        DevCipherVector o = rescaleDev(x, cps->scale, qi());
        if (o.level != x.level - 1) throw std::runtime_error("EvaluateCheby: a Rescale of the recursion did not divide by exactly one modulus (scale out of range)");
        return o;
    }
    DevCipherVector recurse(double targetScale, uint64_t logSplit, uint64_t logDegree, const ChebyPoly &p) {
        if (p.Degree() < (1ULL << logSplit)) {
            if (resplit(p, logSplit)) { const uint64_t ld = bitLen(p.Degree()); return recurse(targetScale, ld >> 1, ld, p); }
            return leaf(targetScale, p);
        }
        const uint64_t np = nextPower(p, logSplit);
        ChebyPoly q, r; splitCoeffsCheby(p, np, q, r);
        int lm = 0; outLevel(p, logSplit, logDegree, &lm);
        const DevCipherVector &T = C.at(np);
        DevCipherVector res = recurse(targetScale * (double)qi()[lm] / T.scale, logSplit, logDegree, q);
        DevCipherVector tmp = recurse(targetScale, logSplit, logDegree, r);
        if (res.level > tmp.level && res.level != tmp.level + 1) res = DropLevelDev(res, tmp.level + 1);
        res = mulRelinDev(cps, res, T);
        if (res.level != lm) throw std::runtime_error("EvaluateCheby: level prediction of the dry run failed");
        if (res.level > tmp.level) return CAddSubDev(cps, rescaleOnce(res), tmp, false);
        return rescaleOnce(CAddSubDev(cps, res, tmp, false));
    }
};
}  // namespace detail
// eval.EvaluateCheby(op, cheby, targetScale): op must already carry the change of variable 2/(b-a) x + (-a-b)/(b-a)
inline DevCipherVector EvaluateChebyDev(CryptoParams *cps, const DevCipherVector &op, const ChebyshevInterpolation &cheby, double targetScale) {
    const uint64_t logDegree = detail::bitLen(cheby.poly.Degree()), logSplit = logDegree >> 1;
    if (op.level < (int)std::ceil(std::log2((double)(cheby.poly.Degree() + 1))) + 1) throw std::runtime_error("EvaluateCheby: not enough levels");
    detail::ChebyEval ev{cps, {}};
    ev.C[1] = op;
    for (uint64_t i = 2; i < (1ULL << logSplit); i++) ev.powerBasis(i);
    for (uint64_t i = logSplit; i < logDegree; i++) ev.powerBasis(1ULL << i);
    return ev.recurse(targetScale, logSplit, logDegree, cheby.poly);
}
// crypto.ChebyApproximation (basics.go:613-626): EvaluateCheby at the ciphertexts' own scale
inline DevCipherVector ChebyApproximationDev(CryptoParams *cps, const DevCipherVector &X, const ChebyshevInterpolation &cheby) { return EvaluateChebyDev(cps, X, cheby, X.scale); }
// mpc.SigmoidApprox / CSigmoidApprox without their network step (mhe.go:634-667): the caller bootstraps first when SigmoidNeedsBootstrap says so (:641-645)
inline bool SigmoidNeedsBootstrap(int level, int degree) { return level < (int)std::ceil(std::log2((double)(degree + 1)) + 1) + 2; }
inline DevCipherVector CSigmoidApproxLocal(CryptoParams *cps, const DevCipherVector &ctIn, double A, double B, int degree) {
    if (degree == 0) throw std::runtime_error("CSigmoidApprox: Degree == 0 is the decrypt-and-recompute debugging branch (mhe.go:621-632), not a device computation");
    const ChebyshevInterpolation cheby = Approximate(Sigmoid, A, B, degree);
    DevCipherVector y = detail::rescaleDev(CMultConstDev(cps, ctIn, 2 / (B - A)), cps->scale, cps->qi);      // MultByConstNew + Rescale(params.Scale())
    y = AddConstDev(cps, y, (-A - B) / (B - A), cps->qi);
    return EvaluateChebyDev(cps, y, cheby, y.scale);
}

// ---- vectors whose ciphertexts keep their OWN level and scale (a crypto.CipherVector is []*ckks.Ciphertext: NetDQRenc's Householder vector has one
// ciphertext - the one that received alphaScaled - below the others, qrfact.go:166-169).  Element-wise ops act pairwise at the lower level of each pair.
using CellVec = std::vector<DevCipherVector>;                           // every element: n == 1
inline CellVec cellsOf(const DevCipherVector &v) { CellVec c; for (size_t j = 0; j < v.n; j++) c.push_back(viewOne(v, j)); return c; }
inline CellVec CMultCells(CryptoParams *cps, const CellVec &X, const CellVec &Y) {      // crypto.CMult with its length-1 broadcast (basics.go:386-427)
    const size_t n = std::max(X.size(), Y.size()); CellVec o;
    if ((X.size() != n && X.size() != 1) || (Y.size() != n && Y.size() != 1)) throw std::runtime_error("CMult: vector lengths differ");
    for (size_t k = 0; k < n; k++) o.push_back(CMultDev(cps, X[X.size() == 1 ? 0 : k], Y[Y.size() == 1 ? 0 : k], cps->qi));
    return o;
}
inline CellVec CAddCells(CryptoParams *cps, const CellVec &X, const CellVec &Y) {       // crypto.CAdd (basics.go:568-578)
    if (X.size() != Y.size()) throw std::runtime_error("CAdd: vector lengths differ");
    CellVec o; for (size_t k = 0; k < X.size(); k++) o.push_back(CAddSubDev(cps, X[k], Y[k], false));
    return o;
}
// crypto.InnerSumAll (basics.go:278-292): vecsum = X[0]; eval.Add(X[i], vecsum, vecsum) for i >= 1 (each at the lower level of the pair, with lattigo's scale
// matching; an unmatched pair keeps vecsum's scale), then 13 rotate-and-add steps.  (That a binary op yields the LOWER level is this restatement's reading
// of lattigo's evaluateInPlace - parity unpinned.)
inline DevCipherVector InnerSumAllCells(CryptoParams *cps, const CellVec &X) {
    DevCipherVector vecsum = X[0];
    for (size_t k = 1; k < X.size(); k++) vecsum = CAddSubDev(cps, vecsum, X[k], false);
    return InnerSumAllDev(cps, vecsum);
}
inline CipherVector ToHost(const CellVec &c) { CipherVector out; for (const auto &v : c) { CipherVector h = ToHost(v); out.insert(out.end(), h.begin(), h.end()); } return out; }
}  // namespace crypto

namespace gwas {

// ---------------------------------------------------------------- GenoFileStream (filestream.go:284-494)
class GenoFileStream {
    std::string filename; FILE *file = nullptr;
    uint64_t numRows, numCols, lineCount = 0;
    std::vector<uint8_t> buf;
    std::vector<bool> filtRows, filtCols; bool hasRowFilt = false, hasColFilt = false;
    uint64_t filtNumRow = 0, filtNumCol = 0;
    bool replaceMissing;
    std::vector<int8_t> readRow() {                                  // :327-360
        if (CheckEOF()) return {};
        if (fread(buf.data(), 1, numCols, file) != numCols) throw std::runtime_error("GenoFileStream: short read");   // panic(err)
        std::vector<int8_t> out; out.reserve(hasColFilt ? filtNumCol : numCols);
        for (uint64_t i = 0; i < numCols; i++) if (!hasColFilt || filtCols[i]) {
            int8_t v = (int8_t)buf[i];
            if (replaceMissing && v < 0) v = 0;                     // :352-354
            out.push_back(v);
        }
        lineCount++;
        return out;
    }
public:
    GenoFileStream(const std::string &fn, uint64_t numRow, uint64_t numCol, bool replaceMissing_)   // NewGenoFileStream :302-325
        : filename(fn), numRows(numRow), numCols(numCol), buf(numCol), replaceMissing(replaceMissing_) {
        file = fopen(fn.c_str(), "rb");
        if (!file) throw std::runtime_error("NewGenoFileStream: cannot open " + fn);
    }
    ~GenoFileStream() { if (file) fclose(file); }
    void Reset() {                                                   // :362-376
        if (!file) file = fopen(filename.c_str(), "rb"); else fseek(file, 0, SEEK_SET);
        if (!file) throw std::runtime_error("GenoFileStream.Reset: cannot open " + filename);
        lineCount = 0;
    }
    uint64_t NumRows() const { return numRows; }
    uint64_t NumCols() const { return numCols; }
    uint64_t NumRowsToKeep() const { return hasRowFilt ? filtNumRow : numRows; }   // :386-391
    uint64_t NumColsToKeep() const { return hasColFilt ? filtNumCol : numCols; }   // :393-398
    bool CheckEOF() {                                                // :400-412
        if (lineCount >= numRows) { if (file) fclose(file); file = nullptr; return true; }
        return false;
    }
    std::vector<int8_t> NextRow() {                                  // :414-426
        if (CheckEOF()) return {};
        if (hasRowFilt) while (lineCount < filtRows.size() && !filtRows[lineCount]) readRow();
        return readRow();
    }
    int UpdateRowFilt(const std::vector<bool> &a) {                  // :428-454
        if (a.size() != NumRowsToKeep()) throw std::runtime_error("Invalid length of input array");
        if (!hasRowFilt) { filtRows.assign(numRows, true); hasRowFilt = true; }
        int sum = 0; size_t idx = 0;
        for (size_t i = 0; i < filtRows.size(); i++) if (filtRows[i]) { filtRows[i] = a[idx++]; if (filtRows[i]) sum++; }
        filtNumRow = sum; return sum;
    }
    int UpdateColFilt(const std::vector<bool> &a) {                  // :456-482
        if (a.size() != NumColsToKeep()) throw std::runtime_error("Invalid length of input array");
        if (!hasColFilt) { filtCols.assign(numCols, true); hasColFilt = true; }
        int sum = 0; size_t idx = 0;
        for (size_t i = 0; i < filtCols.size(); i++) if (filtCols[i]) { filtCols[i] = a[idx++]; if (filtCols[i]) sum++; }
        filtNumCol = sum; return sum;
    }
    uint64_t LineCount() const { return lineCount; }
};

// ---------------------------------------------------------------- DiagCacheStream (filestream.go:19-282)
// header: 6 x u64 LE {vectorLen, level, scale bits, n, numModuli, rowSize}, d baby flags, d giant flags;
// record: u64 LE length, u32 LE shift, per plaintext u8 isEmpty + numModuli*n coefficients (big-endian u64,
// lattigo ring.WriteCoeffsTo — unverified).  Plaintexts are NTT + Montgomery form in the file.
struct PlainVector { std::vector<std::vector<uint64_t>> pt; std::vector<bool> empty; };
class DiagCacheStream {
    FILE *file = nullptr; bool isWrite, atHead = true; int d;
    std::vector<uint8_t> buf;
public:
    uint64_t vectorLen = 0, level = 0, n = 0, numModuli = 0, rowSize = 0; double scale = 0;
    std::vector<bool> babyTable, giantTable;
    static std::string FileName(const std::string &prefix, int blockRowIndex) { return prefix + "_" + std::to_string(blockRowIndex) + ".bin"; }   // :43
    // second result of NewDiagCacheStream: true when a write was requested but the file already exists (:48-54)
    DiagCacheStream(const std::string &prefix, int blockRowIndex, bool isWrite_, int slots, bool *existed = nullptr) : isWrite(isWrite_) {
        d = (int)std::ceil(std::sqrt((double)slots));
        const std::string fn = FileName(prefix, blockRowIndex);
        if (existed) *existed = false;
        if (isWrite) {
            if (FILE *t = fopen(fn.c_str(), "rb")) { fclose(t); if (existed) *existed = true; return; }
            file = fopen(fn.c_str(), "wb");
        } else file = fopen(fn.c_str(), "rb");
        if (!file) throw std::runtime_error("NewDiagCacheStream: cannot open " + fn);
        if (!isWrite) {
            uint8_t h[48];
            if (fread(h, 1, 48, file) != 48) throw std::runtime_error("DiagCacheStream: short header");
            auto le = [&](int k) { uint64_t v = 0; for (int i = 0; i < 8; i++) v |= (uint64_t)h[8 * k + i] << (8 * i); return v; };
            vectorLen = le(0); level = le(1); uint64_t sb = le(2); memcpy(&scale, &sb, 8); n = le(3); numModuli = le(4); rowSize = le(5);
            std::vector<uint8_t> tb(2 * d);
            if (fread(tb.data(), 1, 2 * d, file) != (size_t)(2 * d)) throw std::runtime_error("DiagCacheStream: short tables");
            babyTable.resize(d); giantTable.resize(d);
            for (int i = 0; i < d; i++) { babyTable[i] = tb[i] != 0; giantTable[i] = tb[d + i] != 0; }
            buf.resize(rowSize);
        }
    }
    ~DiagCacheStream() { Close(); }
    void SetIndexTables(const std::vector<bool> &b, const std::vector<bool> &g) { babyTable = b; giantTable = g; }   // :134-137
    void WriteDiag(const PlainVector &pv, uint32_t shift, uint64_t level_, double scale_, uint64_t n_, uint64_t numModuli_) {   // :144-231
        if (atHead) {
            vectorLen = pv.pt.size(); level = level_; scale = scale_; n = n_; numModuli = numModuli_;
            rowSize = 4 + (1 + n * numModuli * 8) * vectorLen;
            if (babyTable.empty() || giantTable.empty()) throw std::runtime_error("babyTable or giantTable not set before attempting to write diag cache header");
            uint8_t h[48]; uint64_t sb; memcpy(&sb, &scale, 8);
            uint64_t f[6] = {vectorLen, level, sb, n, numModuli, rowSize};
            for (int k = 0; k < 6; k++) for (int i = 0; i < 8; i++) h[8 * k + i] = (uint8_t)(f[k] >> (8 * i));
            fwrite(h, 1, 48, file);
            for (bool v : babyTable) fputc(v ? 1 : 0, file);
            for (bool v : giantTable) fputc(v ? 1 : 0, file);
            buf.resize(rowSize); atHead = false;
        }
        size_t ptr = 0;
        for (int i = 0; i < 4; i++) buf[ptr++] = (uint8_t)(shift >> (8 * i));
        for (size_t i = 0; i < pv.pt.size(); i++) {
            buf[ptr++] = pv.empty[i] ? 1 : 0;
            if (!pv.empty[i]) for (uint64_t w : pv.pt[i]) for (int b = 7; b >= 0; b--) buf[ptr++] = (uint8_t)(w >> (8 * b));
        }
        uint8_t l8[8]; for (int i = 0; i < 8; i++) l8[i] = (uint8_t)((uint64_t)ptr >> (8 * i));
        fwrite(l8, 1, 8, file); fwrite(buf.data(), 1, ptr, file);
    }
    bool ReadDiag(PlainVector &pv, int &shift) {                     // :247-282; false at EOF (the Go version returns nil)
        uint8_t l8[8];
        if (!file || fread(l8, 1, 8, file) != 8) return false;
        uint64_t len = 0; for (int i = 0; i < 8; i++) len |= (uint64_t)l8[i] << (8 * i);
        if (len > rowSize || fread(buf.data(), 1, len, file) != len) return false;
        shift = (int)((uint32_t)buf[0] | (uint32_t)buf[1] << 8 | (uint32_t)buf[2] << 16 | (uint32_t)buf[3] << 24);
        size_t ptr = 4; pv.pt.assign(vectorLen, {}); pv.empty.assign(vectorLen, false);
        for (uint64_t i = 0; i < vectorLen; i++) {
            pv.empty[i] = buf[ptr++] == 1;
            if (!pv.empty[i]) { pv.pt[i].resize(numModuli * n); for (auto &w : pv.pt[i]) { w = 0; for (int b = 0; b < 8; b++) w = (w << 8) | buf[ptr++]; } }
        }
        return true;
    }
    void Close() { if (file) { fclose(file); file = nullptr; } }
};

// ---------------------------------------------------------------- the products
inline std::vector<uint64_t> flattenCipherMatrix(const crypto::CipherMatrix &A) {
    std::vector<uint64_t> f;
    for (auto &row : A) for (auto &ct : row) f.insert(f.end(), ct.data.begin(), ct.data.end());
    return f;
}
inline crypto::CipherMatrix unflatten(const std::vector<uint64_t> &f, int s, int m_ct, int level, double scale, int N) {
    crypto::CipherMatrix out(s, crypto::CipherVector(m_ct));
    const size_t w = (size_t)2 * (level + 1) * N;
    for (int i = 0; i < s; i++) for (int j = 0; j < m_ct; j++) {
        auto &ct = out[i][j]; ct.level = level; ct.scale = scale;
        ct.data.assign(f.begin() + ((size_t)i * m_ct + j) * w, f.begin() + ((size_t)i * m_ct + j + 1) * w);
    }
    return out;
}
inline std::vector<int8_t> readAllRows(GenoFileStream *gfs, uint64_t &nrow, uint64_t &ncol) {
    gfs->Reset();                                                    // matmult.go:1239
    nrow = gfs->NumRowsToKeep(); ncol = gfs->NumColsToKeep();
    std::vector<int8_t> geno(nrow * ncol);
    for (uint64_t r = 0; r < nrow; r++) { auto row = gfs->NextRow(); if (row.size() != ncol) throw std::runtime_error("GenoFileStream: unexpected row length"); memcpy(&geno[r * ncol], row.data(), ncol); }
    return geno;
}

// matmult.go:1238.  Returns (out, sum, sqSum); sum/sqSum are empty unless computeSquaredSum.  The returned
// ciphertexts are the deterministic product; the reference adds them onto CZeroMat (a fresh encryption of zero).
inline std::tuple<crypto::CipherMatrix, std::vector<double>, std::vector<double>>
MatMult4Stream(crypto::CryptoParams *cps, const crypto::CipherMatrix &A, GenoFileStream *gfs, int maxLevel, bool computeSquaredSum, bool square, int /*nproc*/) {
    uint64_t nrow, ncol;
    std::vector<int8_t> geno = readAllRows(gfs, nrow, ncol);
    const int s = (int)A.size(), inLevel = A[0][0].Level(), slots = cps->GetSlots();
    const double outScale = A[0][0].Scale() * cps->scale;            // :1247
    const int m_ct = (int)((ncol - 1) / slots) + 1, numBlockRows = (int)((nrow - 1) / slots) + 1;   // :1253-1254
    if ((int)A[0].size() != numBlockRows) throw std::runtime_error("MatMult4Stream: A has the wrong number of block rows");
    std::vector<uint64_t> a = flattenCipherMatrix(A), o((size_t)s * m_ct * 2 * maxLevel * cps->N());
    std::vector<double> sum, sq;
    if (computeSquaredSum) { sum.assign(ncol, 0); sq.assign(ncol, 0); }
    cps->check(sfg_matmul_stream(cps->ctx, a.data(), s, inLevel, maxLevel, geno.data(), nrow, ncol, ncol, square ? SFG_SQUARE : 0u, o.data(),
                                 computeSquaredSum ? sum.data() : nullptr, computeSquaredSum ? sq.data() : nullptr), "MatMult4Stream");
    return {unflatten(o, s, m_ct, maxLevel - 1, outScale, cps->N()), sum, sq};
}

// matmult.go:914 / :1043.  The DiagCache files of the reference become an HBM-resident int8 matrix keyed by the
// same cacheFilePrefix; `transposeOf` registers a prefix as the transpose of an already resident matrix, which is how
// pca.go:112-113 (X cache, X^T cache) maps onto ONE resident copy.
// The matrix is read the way the reference reads it - one row at a time out of GenoFileStream::NextRow (matmult.go:942-950, filestream.go:414-426) - into a
// staging buffer of a few thousand rows (kPreprocessStagingBytes) that sfg_geno_write_rows hands to the device: nothing here scales with nrow * ncol.  A prefix whose
// matrix is the TRANSPOSE of a resident one is recognised on the device, chunk by chunk (sfg_geno_compare_rows: exact comparison, X^T is never held), and becomes a
// view of the one int8 copy - the explicit `transposeOf` says so without reading the file.
constexpr size_t kPreprocessStagingBytes = 64u << 20;
template <class Visit> inline void streamRows(GenoFileStream *gfs, uint64_t nrow, uint64_t ncol, Visit visit, size_t stagingBytes = kPreprocessStagingBytes) {
    gfs->Reset();
    const uint64_t per = std::max<uint64_t>(1, stagingBytes / ncol);
    std::vector<int8_t> stage(per * ncol);
    for (uint64_t row0 = 0; row0 < nrow; row0 += per) {
        const uint64_t n = std::min(per, nrow - row0);
        for (uint64_t r = 0; r < n; r++) { auto row = gfs->NextRow(); if (row.size() != ncol) throw std::runtime_error("GenoFileStream: unexpected row length"); memcpy(&stage[r * ncol], row.data(), ncol); }
        if (!visit(row0, n, stage.data())) return;
    }
}
inline void MatMult4StreamPreprocess(crypto::CryptoParams *cps, GenoFileStream *gfs, int /*maxLevel*/, const std::string &cacheFilePrefix,
                                     const std::string &transposeOf = "", size_t stagingBytes = kPreprocessStagingBytes) {
    std::lock_guard<std::mutex> lk(cps->resident->mu);
    auto &tab = cps->resident->tab;
    if (tab.count(cacheFilePrefix)) return;                          // "Found cache file" (filestream.go:52-54): skip
    if (!transposeOf.empty()) { auto it = tab.find(transposeOf); if (it == tab.end()) throw std::runtime_error("transposeOf prefix is not resident"); tab[cacheFilePrefix] = {it->second.g, SFG_TRANSPOSE, false, it->second.mg, it->second.nrow, it->second.ncol}; return; }
    const uint64_t nrow = gfs->NumRowsToKeep(), ncol = gfs->NumColsToKeep();
    sfg_mgpu *eng = cps->mg ? cps->mg : cps->mg_root;
    auto fail = [&](const char *what) { throw std::runtime_error(std::string("MatMult4StreamPreprocess: ") + what + ": " + (eng ? sfg_mgpu_last_error(eng) : sfg_last_error(cps->ctx))); };
    for (auto &kv : tab) {                                           // pca.go:113: is this the transpose of a matrix that is already resident?
        const crypto::ResidentGeno &r = kv.second;
        if (r.flags || r.nrow != ncol || r.ncol != nrow) continue;
        uint64_t ndiff = 0;
        streamRows(gfs, nrow, ncol, [&](uint64_t row0, uint64_t n, const int8_t *chunk) {
            if (r.mg ? sfg_mgpu_geno_compare_rows(eng, r.mg, SFG_TRANSPOSE, row0, n, chunk, ncol, &ndiff) : sfg_geno_compare_rows(cps->ctx, r.g, SFG_TRANSPOSE, row0, n, chunk, ncol, &ndiff)) fail("compare_rows");
            return ndiff == 0;
        }, stagingBytes);
        if (!ndiff) { tab[cacheFilePrefix] = {r.g, SFG_TRANSPOSE, false, r.mg, r.nrow, r.ncol}; return; }
    }
    crypto::ResidentGeno r; r.nrow = nrow; r.ncol = ncol;
    if (cps->mg) {               // the STORED orientation's columns (pca.go:112: X, individuals x SNPs) are the shards
        if (sfg_mgpu_geno_create(cps->mg, nrow, ncol, &r.mg)) fail("geno_create");
    } else cps->check(sfg_geno_create(cps->ctx, nrow, ncol, &r.g), "MatMult4StreamPreprocess");
    r.owner = true;
    try {
        streamRows(gfs, nrow, ncol, [&](uint64_t row0, uint64_t n, const int8_t *chunk) {
            if (r.mg ? sfg_mgpu_geno_write_rows(cps->mg, r.mg, row0, n, chunk, ncol) : sfg_geno_write_rows(cps->ctx, r.g, row0, n, chunk, ncol)) fail("write_rows");
            return true;
        }, stagingBytes);
    } catch (...) { if (r.mg) sfg_mgpu_geno_free(cps->mg, r.mg); else sfg_geno_free(cps->ctx, r.g); throw; }
    tab[cacheFilePrefix] = r;
}
inline crypto::CipherMatrix MatMult4StreamCompute(crypto::CryptoParams *cps, const crypto::CipherMatrix &A, int maxLevel, const std::string &cacheFilePrefix,
                                                  int m_ct) {
    crypto::ResidentGeno rg;
    {
        std::lock_guard<std::mutex> lk(cps->resident->mu);
        auto it = cps->resident->tab.find(cacheFilePrefix);
        if (it != cps->resident->tab.end()) rg = it->second;
    }
    if (rg.mg) {                 // every GPU of the node: shards, the exchange of Q' X^T and the gather happen inside the library (mgpu.hip)
        const int s = (int)A.size(), inLevel = A[0][0].Level();
        std::vector<uint64_t> a = flattenCipherMatrix(A), o((size_t)s * m_ct * 2 * maxLevel * cps->N());
        sfg_mgpu *eng = cps->mg ? cps->mg : cps->mg_root;              // a fork multiplies on the engine of its root (one call at a time per engine: the caller's rule, as for any context)
        if (!eng) throw std::runtime_error("MatMult4StreamCompute: the matrix of prefix " + cacheFilePrefix + " is sharded over a multi-GPU engine this CryptoParams has no access to");
        if (sfg_mgpu_matmul(eng, a.data(), s, inLevel, maxLevel, rg.mg, rg.flags, o.data())) throw std::runtime_error(std::string("MatMult4StreamCompute: ") + sfg_mgpu_last_error(eng));
        return unflatten(o, s, m_ct, maxLevel - 1, A[0][0].Scale() * cps->scale, cps->N());
    }
    if (!rg.g) {
        // not resident: an on-disk DiagCache written by the reference (or by a CPU-only party) under this prefix is consumed as is
        FILE *probe = fopen(DiagCacheStream::FileName(cacheFilePrefix, 0).c_str(), "rb");
        if (!probe) throw std::runtime_error("MatMult4StreamCompute: no resident matrix and no cache file for prefix " + cacheFilePrefix);   // os.Open panics (filestream.go:59-61)
        fclose(probe);
        const int s = (int)A.size(), inLevel = A[0][0].Level(), nbr = (int)A[0].size();
        std::vector<uint64_t> a = flattenCipherMatrix(A), o((size_t)s * m_ct * 2 * maxLevel * cps->N());
        crypto::detail::DevBuf dA(cps, a.size() * 8), dO(cps, o.size() * 8);
        cps->check(sfg_memcpy_h2d(cps->ctx, dA.p, a.data(), a.size() * 8), "MatMult4StreamCompute");
        cps->check(sfg_matmul_from_cache(cps->ctx, dA.u(), s, inLevel, maxLevel, cacheFilePrefix.c_str(), nbr, dO.u()), "MatMult4StreamCompute");
        cps->check(sfg_memcpy_d2h(cps->ctx, o.data(), dO.p, o.size() * 8), "MatMult4StreamCompute");
        return unflatten(o, s, m_ct, maxLevel - 1, A[0][0].Scale() * cps->scale, cps->N());
    }
    const int s = (int)A.size(), inLevel = A[0][0].Level();
    std::vector<uint64_t> a = flattenCipherMatrix(A), o((size_t)s * m_ct * 2 * maxLevel * cps->N());
    void *dA = nullptr, *dO = nullptr;
    cps->check(sfg_malloc(cps->ctx, &dA, a.size() * 8), "MatMult4StreamCompute"); cps->check(sfg_malloc(cps->ctx, &dO, o.size() * 8), "MatMult4StreamCompute");
    cps->check(sfg_memcpy_h2d(cps->ctx, dA, a.data(), a.size() * 8), "MatMult4StreamCompute");
    int rc = sfg_matmul_resident_dev(cps->ctx, (const uint64_t *)dA, s, inLevel, maxLevel, rg.g, rg.flags, (uint64_t *)dO);
    if (!rc) rc = sfg_memcpy_d2h(cps->ctx, o.data(), dO, o.size() * 8);
    sfg_free(cps->ctx, dA); sfg_free(cps->ctx, dO);
    cps->check(rc, "MatMult4StreamCompute");
    return unflatten(o, s, m_ct, maxLevel - 1, A[0][0].Scale() * cps->scale, cps->N());
}


// ---------------------------------------------------------------- device-resident forms of the product and of its wrappers
// MatMult4StreamCompute on a resident matrix (matmult.go:1043): A [s][nbr] resident, result [s][m_ct] resident at level maxLevel-1
inline crypto::DevCipherMatrix MatMult4StreamComputeDev(crypto::CryptoParams *cps, const crypto::DevCipherMatrix &A, int maxLevel, const std::string &cacheFilePrefix, int m_ct) {
    crypto::ResidentGeno rg;
    { std::lock_guard<std::mutex> lk(cps->resident->mu); auto it = cps->resident->tab.find(cacheFilePrefix);
      if (it == cps->resident->tab.end()) throw std::runtime_error("MatMult4StreamCompute: no resident matrix for prefix " + cacheFilePrefix); rg = it->second; }
    if (rg.mg) throw std::runtime_error("MatMult4StreamComputeDev: the matrix is sharded over several GPUs - use MatMult4StreamCompute (host ciphertexts in, host ciphertexts out)");
    crypto::DevCipherMatrix out = crypto::NewDevCipherMatrix(cps, A.rows, (size_t)m_ct, maxLevel - 1, A.scale * cps->scale);
    cps->check(sfg_matmul_resident_dev(cps->ctx, A.buf->u(), (int)A.rows, A.level, maxLevel, rg.g, rg.flags, out.buf->u()), "MatMult4StreamCompute");
    return out;
}
// QXLazyNormStream (matmult.go:27-77) = local part 1, BootstrapMatAll (network, stays in Go), local part 2.
//   part 1: QS[i] = CMult(Q[i], XStdInv);  out = MatMult4StreamCompute(QS, 5, Xcache)                         (:36-42)
//   part 2: QSm[i] = InnerProd(QS[i], XMean);  out[i][j] = MaskTrunc(out[i][j] - QSm[i], N_j)                 (:47-71)
struct QXLazyNormState { crypto::DevCipherMatrix QS; };
inline crypto::DevCipherMatrix QXLazyNormStreamLocal1(crypto::CryptoParams *cps, const crypto::DevCipherMatrix &Q, const std::string &Xcachefile, int m_ct,
                                                      const crypto::DevCipherVector &XStdInv, const std::vector<uint64_t> &qi, QXLazyNormState &st) {
    crypto::DevCipherVector first = crypto::CMultDev(cps, Q.row(0), XStdInv, qi);
    st.QS = crypto::NewDevCipherMatrix(cps, Q.rows, Q.cols, first.level, first.scale);
    const size_t roww = Q.cols * crypto::detail::ctWords(cps, first.level);
    for (size_t i = 0; i < Q.rows; i++) {
        crypto::DevCipherVector r = i ? crypto::CMultDev(cps, Q.row(i), XStdInv, qi) : first;
        cps->check(sfg_memcpy_d2d(cps->ctx, st.QS.row(i).ptr(), r.ptr(), roww * 8), "d2d");
    }
    return MatMult4StreamComputeDev(cps, st.QS, 5, Xcachefile, m_ct);
}
// The result keeps PER-CIPHERTEXT level and scale, as the reference's [][]*ckks.Ciphertext does: MaskTrunc (basics.go:110-127) returns a full-slot
// column untouched (level l, scale S) and multiplies + rescales only the ragged last column (level l-1, scale S * Delta / q_l), matmult.go:60-70.
inline crypto::DevCipherCells QXLazyNormStreamLocal2(crypto::CryptoParams *cps, const crypto::DevCipherMatrix &outBootstrapped, const QXLazyNormState &st,
                                                     const crypto::DevCipherVector &XMean, int numInd, const std::vector<uint64_t> &qi) {
    const int slots = cps->GetSlots();
    crypto::DevCipherCells cells(outBootstrapped.rows);
    for (size_t i = 0; i < outBootstrapped.rows; i++) {
        crypto::DevCipherVector QSm = crypto::InnerProdDev(cps, st.QS.row(i), XMean, qi);                     // value in all slots
        crypto::DevCipherVector d = crypto::CAddSubDev(cps, outBootstrapped.row(i), QSm, true);              // eval.Sub(out[i][j], QSm[i], out[i][j])
        for (size_t j = 0; j < d.n; j++) {
            const int Nk = j + 1 < d.n ? slots : ((numInd - 1) % slots) + 1;
            crypto::DevCipherVector one = d; one.off = d.off + j * crypto::detail::ctWords(cps, d.level); one.n = 1;
            cells[i].push_back(crypto::MaskTruncDev(cps, one, Nk, qi));
        }
    }
    return cells;
}
// QXtLazyNormStream (matmult.go:83-116): part 1 = the product (:91), part 2 after the bootstrap (:95-111):
//   out[i][j] = CMult(out[i][j] - CMultScalar(XMean, InnerSumAll(Q[i]))[j], XStdInv[j])
inline crypto::DevCipherMatrix QXtLazyNormStreamLocal2(crypto::CryptoParams *cps, const crypto::DevCipherMatrix &outBootstrapped, const crypto::DevCipherMatrix &Q,
                                                       const crypto::DevCipherVector &XMean, const crypto::DevCipherVector &XStdInv, const std::vector<uint64_t> &qi) {
    std::vector<crypto::DevCipherVector> rows;
    for (size_t i = 0; i < outBootstrapped.rows; i++) {
        crypto::DevCipherVector rowSum = crypto::InnerSumAllDev(cps, Q.row(i));
        crypto::DevCipherVector Q1m = crypto::CMultScalarDev(cps, XMean, rowSum, qi);
        crypto::DevCipherVector d = crypto::CAddSubDev(cps, outBootstrapped.row(i), Q1m, true);
        rows.push_back(crypto::CMultDev(cps, d, XStdInv, qi));
    }
    crypto::DevCipherMatrix out = crypto::NewDevCipherMatrix(cps, rows.size(), rows[0].n, rows[0].level, rows[0].scale);
    for (size_t i = 0; i < rows.size(); i++) cps->check(sfg_memcpy_d2d(cps->ctx, out.row(i).ptr(), rows[i].ptr(), rows[i].n * crypto::detail::ctWords(cps, rows[i].level) * 8), "d2d");
    return out;
}
// DCMatMulAAtB (matmult.go:121-156), column c of A: local part 1 = cTQloc[j] = InnerSumAll(innerFn(A[c], B, j)) with innerFn = CMult
// (the inner function qrfact.go / assoc.go pass), AggregateCVec (network, stays in Go), local part 2 = out[j] += CMult(A[c], {cTQ[j]})
inline crypto::DevCipherVector DCMatMulAAtBLocal1(crypto::CryptoParams *cps, const crypto::DevCipherVector &Ac, const crypto::DevCipherMatrix &B, const std::vector<uint64_t> &qi) {
    std::vector<crypto::DevCipherVector> parts;
    for (size_t j = 0; j < B.rows; j++) parts.push_back(crypto::InnerSumAllDev(cps, crypto::CMultDev(cps, Ac, B.row(j), qi)));
    crypto::DevCipherVector out = crypto::NewDevCipherVector(cps, parts.size(), parts[0].level, parts[0].scale);
    for (size_t j = 0; j < parts.size(); j++) cps->check(sfg_memcpy_d2d(cps->ctx, out.ptr(j), parts[j].ptr(), crypto::detail::ctWords(cps, out.level) * 8), "d2d");
    return out;
}
inline void DCMatMulAAtBLocal2(crypto::CryptoParams *cps, const crypto::DevCipherVector &Ac, const crypto::DevCipherVector &cTQ, std::vector<crypto::DevCipherVector> &out,
                               const std::vector<uint64_t> &qi) {
    for (size_t j = 0; j < cTQ.n; j++) {
        crypto::DevCipherVector one = cTQ; one.off = cTQ.off + j * crypto::detail::ctWords(cps, cTQ.level); one.n = 1;
        crypto::DevCipherVector ccTQ = crypto::CMultDev(cps, Ac, one, qi);
        out[j] = out[j].n ? crypto::CAddSubDev(cps, out[j], ccTQ, false) : ccTQ;                             // out starts as CZeroMat (fresh zero encryptions, added by the Go side)
    }
}

// ---------------------------------------------------------------- f-2: NetDQRenc (gwas/qrfact.go:47-316), the LOCAL segments of one column step
// Between them sit MPC rounds that stay in Go (AggregateSharesCT, CiphertextToSS, SqrtAndSqrtInverse, IsPositive, SSMultElemVec, Trunc, SStoCiphertext,
// AggregateCVec, BootstrapMatAll).  Forward column (qrfact.go:75-216):
//   F1  zloc = SqSum(A[0]); uvec = copy of A[0]                                                                  (:91-93)
//   F2  alphaScaled = Mask(Rebalance(alphaScaled), slotid, false); zNewSqrtInv = Rebalance(zNewSqrtInv)          (:136-141)
//       uvec = CMultScalar(uvec, zNewSqrtInv); the owner of the pivot row: uvec[ctid] += Mask(alphaScaled, slotid, false)   (:159-165)
//   F3  DCMatMulAAtB(vMat = {uvec}, A) = cTQloc[j] = InnerSumAll(CMult(uvec, A[j])) | AggregateCVec | out[j] = CMult(uvec, {cTQ[j]})   (:176-189, matmult.go:121-156)
//       A[c][ci] += vvTA[c][ci] * (-2 / N)   (eval.MultByConstAndAdd, :193-200)
//   F4  after BootstrapMatAll: A = A[1:]; the pivot row is masked out of every remaining column; FlattenLevels                (:202-215)
// Backward column (:236-285): the same DCMatMulAAtB with Mask(vList[j][ctid], slotid, false) as the inner function of the leftmost column, then
//   Q[j + c] += vvTQ[c] * (-2 scalar), scalar = 1/sqrt(N) for c = 0 and 1/N otherwise.
inline crypto::DevCipherVector NetDQRencF1(crypto::CryptoParams *cps, const crypto::CellVec &A0) {
    return crypto::InnerSumAllCells(cps, crypto::CMultCells(cps, A0, A0));                                        // crypto.SqSum = InnerProd(X, X), basics.go:355-358
}
inline crypto::CellVec NetDQRencF2(crypto::CryptoParams *cps, const crypto::CellVec &A0, const crypto::DevCipherVector &alphaScaledIn, const crypto::DevCipherVector &zNewSqrtInvIn,
                                   bool ownsPivot, int ctid, int slotid) {
    crypto::DevCipherVector alphaScaled = crypto::MaskDev(cps, crypto::RebalanceDev(cps, alphaScaledIn), slotid, false, cps->qi);
    crypto::DevCipherVector zNewSqrtInv = crypto::RebalanceDev(cps, zNewSqrtInvIn);
    crypto::CellVec uvec = crypto::CMultCells(cps, A0, crypto::CellVec{zNewSqrtInv});
    if (ownsPivot) {
        alphaScaled = crypto::MaskDev(cps, alphaScaled, slotid, false, cps->qi);
        uvec[ctid] = crypto::CAddSubDev(cps, uvec[ctid], alphaScaled, false);                                     // crypto.Add, basics.go:174-181
    }
    return uvec;
}
// the local half of DCMatMulAAtB before AggregateCVec; pivotMask >= 0: the backward pass's inner function for j = 0 (Mask(a[ctid], slotid, false), or an
// encryption of zero supplied by the Go side when this party does not own the pivot row - pass ownsPivot = false and the result omits entry 0)
inline std::vector<crypto::DevCipherVector> NetDQRencInner(crypto::CryptoParams *cps, const crypto::CellVec &v, const std::vector<crypto::CellVec> &B, bool backward, bool ownsPivot, int ctid, int slotid) {
    std::vector<crypto::DevCipherVector> cTQloc(B.size());
    for (size_t j = 0; j < B.size(); j++) {
        if (backward && j == 0) { if (ownsPivot) cTQloc[0] = crypto::InnerSumAllCells(cps, crypto::CellVec{crypto::MaskDev(cps, v[ctid], slotid, false, cps->qi)}); continue; }
        cTQloc[j] = crypto::InnerSumAllCells(cps, crypto::CMultCells(cps, v, B[j]));
    }
    return cTQloc;
}
// after AggregateCVec: out[j] = CMult(v, {cTQ[j]}) (added onto CZeroMat by the Go side), then M[j] += out[j] * constants[j] (eval.MultByConstAndAdd per ciphertext)
inline void NetDQRencUpdate(crypto::CryptoParams *cps, const crypto::CellVec &v, const std::vector<crypto::DevCipherVector> &cTQ, std::vector<crypto::CellVec> &M, const std::vector<double> &constants) {
    for (size_t j = 0; j < cTQ.size(); j++) {
        crypto::CellVec vv = crypto::CMultCells(cps, v, crypto::CellVec{cTQ[j]});
        for (size_t ci = 0; ci < vv.size(); ci++) crypto::MultByConstAndAddDev(cps, vv[ci], constants[j], M[j][ci], cps->qi);
    }
}
// F4 on the bootstrapped matrix (one level): drop the first column, mask the pivot row out of the pivot owner's ciphertext, FlattenLevels
inline std::vector<crypto::CellVec> NetDQRencF4(crypto::CryptoParams *cps, const std::vector<crypto::CellVec> &Aboot, bool ownsPivot, int ctid, int slotid) {
    std::vector<crypto::CellVec> A(Aboot.begin() + 1, Aboot.end());
    if (ownsPivot) for (auto &col : A) col[ctid] = crypto::MaskDev(cps, col[ctid], slotid, true, cps->qi);
    int lvl = 1 << 30; for (auto &col : A) for (auto &c : col) lvl = std::min(lvl, c.level);
    for (auto &col : A) for (auto &c : col) if (c.level != lvl) c = crypto::DropLevelDev(c, lvl);                // crypto.FlattenLevels, basics.go:514-531
    return A;
}

// ---------------------------------------------------------------- A14: ciphertext x ciphertext matrix helpers of the logistic path
// (matmult.go:1915-2066, called from assoc.go:992-1170).  The reference adds every term onto crypto.InitEncryptedMatrix (fresh
// encryptions of zero); these return the deterministic sums, the Go shim keeps adding them onto its zero matrix.
// CMultMatInnerProd (:1915-1944): result[r][0] = sum_c Mask(InnerProd(M[r], N[c]), c, false)
inline std::vector<crypto::DevCipherVector> CMultMatInnerProdDev(crypto::CryptoParams *cps, const crypto::DevCipherMatrix &M, const crypto::DevCipherMatrix &Nm, const std::vector<uint64_t> &qi) {
    std::vector<crypto::DevCipherVector> result(M.rows);
    for (size_t r = 0; r < M.rows; r++) for (size_t c = 0; c < Nm.rows; c++) {
        crypto::DevCipherVector t = crypto::MaskDev(cps, crypto::InnerProdDev(cps, M.row(r), Nm.row(c), qi), (int)c, false, qi);
        result[r] = result[r].n ? crypto::CAddSubDev(cps, t, result[r], false) : t;
    }
    return result;
}
// CMultMatInnerProdVector (:1947-1986): M rows and N are first multiplied by the 0/1 mask of the first MCols slots; result[0] = sum_k Mask(InnerProd(M[k]*mask, N*mask), k, false)
inline crypto::DevCipherVector CMultMatInnerProdVectorDev(crypto::CryptoParams *cps, const crypto::DevCipherMatrix &M, const crypto::DevCipherVector &Nv, int MCols, const std::vector<uint64_t> &qi) {
    std::vector<double> maskClear((size_t)cps->GetSlots() * M.cols, 0.0);
    for (int i = 0; i < MCols; i++) maskClear[i] = 1.0;
    crypto::DevCipherVector Nmasked = crypto::mulByRealVectorsDev(cps, Nv, maskClear, qi), result;
    for (size_t k = 0; k < M.rows; k++) {
        crypto::DevCipherVector Mcur = crypto::mulByRealVectorsDev(cps, M.row(k), maskClear, qi);
        crypto::DevCipherVector t = crypto::MaskDev(cps, crypto::InnerProdDev(cps, Mcur, Nmasked, qi), (int)k, false, qi);
        result = result.n ? crypto::CAddSubDev(cps, t, result, false) : t;
    }
    return result;
}
// CMultMatColTimesColToCol (:1989-2027) and ...RowToCol (:2030-2066): result[c] = sum_k CMult(replicate(InnerSumAll(CMask(sel, idx, false))), M[k])
// with (sel, idx) = (N[c], k) resp. (N[k], c)
inline std::vector<crypto::DevCipherVector> CMultMatColTimesToColDev(crypto::CryptoParams *cps, const crypto::DevCipherMatrix &M, const crypto::DevCipherMatrix &Nm, int numColsOut, bool rowForm,
                                                                     const std::vector<uint64_t> &qi) {
    std::vector<crypto::DevCipherVector> result(numColsOut);
    for (size_t k = 0; k < M.rows; k++) for (int c = 0; c < numColsOut; c++) {
        crypto::DevCipherVector elemRep = rowForm ? crypto::CMaskDev(cps, Nm.row(k), c, false, qi) : crypto::CMaskDev(cps, Nm.row(c), (int)k, false, qi);
        crypto::DevCipherVector elemRepCiph = crypto::InnerSumAllDev(cps, elemRep);
        crypto::DevCipherVector multi = crypto::CMultDev(cps, elemRepCiph, M.row(k), qi);        // length-1 broadcast over the cts of M[k]
        result[c] = result[c].n ? crypto::CAddSubDev(cps, multi, result[c], false) : multi;
    }
    return result;
}
}  // namespace gwas

namespace mpc {
// mpc_core.RVec flattened: `limbs` little-endian 64-bit words per element (2 = LElem128, 4 = LElem256)
struct RVec { int limbs = 2; std::vector<uint64_t> w; size_t size() const { return w.size() / limbs; } };
struct MPC { crypto::CryptoParams *cps; int pid; std::vector<uint64_t> modulus; };
// mpc/beavermult.go:108-133
inline RVec BeaverMultElemVec(MPC *m, const RVec &ar, const RVec &am, const RVec &br, const RVec &bm) {
    RVec out; out.limbs = am.limbs; out.w.resize(am.w.size());
    m->cps->check(sfg_beaver_elem(m->cps->ctx, m->pid, am.limbs, m->modulus.data(), ar.w.data(), am.w.data(), br.w.data(), bm.w.data(), out.w.data(), am.size()), "BeaverMultElemVec");
    return out;
}
// mpc/beavermult.go:135-147: (m x k) * (k x n)
inline RVec BeaverMultMat(MPC *m, const RVec &ar, const RVec &am, const RVec &br, const RVec &bm, int rows, int inner, int cols) {
    RVec out; out.limbs = am.limbs; out.w.resize((size_t)rows * cols * am.limbs);
    m->cps->check(sfg_beaver_matmul(m->cps->ctx, m->pid, am.limbs, m->modulus.data(), ar.w.data(), am.w.data(), br.w.data(), bm.w.data(), out.w.data(), rows, inner, cols), "BeaverMultMat");
    return out;
}

// ---- collective bootstrap, local halves (mpc/mhe.go:289-348 CollectiveBootstrapMat).  The network steps between them stay in Go:
//      broadcast of cm (:296-311), AggregateRefreshShareMat (:326-327).  PARITY UNPINNED (see include/sfgwas_hip.h).
// Per-ciphertext randomness is drawn by the caller, as the reference draws it (ring.RandInt masks, the Gaussian sampler, crpGen.ReadNew()):
//   masks [nct][N][maskLimbs] two's-complement limbs, e0 / e1 [nct][N], crs [nct][nq][N]  - all device resident
struct RefreshRandomness { const uint64_t *mask = nullptr; int maskLimbs = 0; const int32_t *e0 = nullptr, *e1 = nullptr; const uint64_t *crs = nullptr; };
struct RefreshShares {            // refSharesDecrypt [nct][level+1][N], refSharesRecrypt [nct][nq][N]
    std::shared_ptr<crypto::detail::DevBuf> h0, h1; size_t nct = 0; int level = 0;
};
// mhe.go:313-324: FlattenLevels has already been applied (cm is one level); one GenShares per ciphertext, batched
inline RefreshShares CollectiveBootstrapGenShares(crypto::CryptoParams *cps, const crypto::DevCipherMatrix &cm, const RefreshRandomness &rnd) {
    RefreshShares sh; sh.nct = cm.rows * cm.cols; sh.level = cm.level;
    const size_t N = (size_t)cps->N();
    sh.h0 = std::make_shared<crypto::detail::DevBuf>(cps, sh.nct * (cm.level + 1) * N * 8);
    sh.h1 = std::make_shared<crypto::detail::DevBuf>(cps, sh.nct * cps->nq * N * 8);
    // mhe.go:315: GenShares(skShard, levelStart, nParties, cm[i][j], parameters.Scale(), crp, ...) - the target scale is ALWAYS Params.Scale(), whatever
    // scale the ciphertexts carry (products arrive at A.scale * Delta, matmult.go:44,92)
    cps->check(sfg_refresh_gen_shares_scaled_dev(cps->ctx, cm.buf->u(), (int)sh.nct, cm.level, cm.scale, cps->scale, rnd.crs, rnd.mask, rnd.maskLimbs, rnd.e0, rnd.e1,
                                                 sh.h0->u(), sh.h1->u()), "RefreshProtocol.GenShares");
    return sh;
}
// mhe.go:329-346: Decrypt, Recode(cm[i][j], parameters.Scale()), Recrypt with the aggregated shares; the result is at MaxLevel with scale Params.Scale()
inline crypto::DevCipherMatrix CollectiveBootstrapFinish(crypto::CryptoParams *cps, const crypto::DevCipherMatrix &cm, const uint64_t *h0agg, const uint64_t *h1agg,
                                                         const uint64_t *crs) {
    crypto::DevCipherMatrix out = crypto::NewDevCipherMatrix(cps, cm.rows, cm.cols, cps->nq - 1, cps->scale);
    cps->check(sfg_refresh_finish_scaled_dev(cps->ctx, cm.buf->u(), (int)(cm.rows * cm.cols), cm.level, cm.scale, cps->scale, h0agg, h1agg, crs, out.buf->u()),
               "RefreshProtocol.Decrypt/Recode/Recrypt");
    return out;
}
}  // namespace mpc
