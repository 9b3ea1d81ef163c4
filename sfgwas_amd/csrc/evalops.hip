// evalops.hip — the remaining ckks.Evaluator operations the PCA / QR / association callers apply between the
// matrix products, batched over ciphertexts that stay in HBM:
//   crypto.CMult  = MulRelinNew + Rescale      (crypto/basics.go:386-427, :229)
//   crypto.CPMult / Mask = ct x plaintext + Rescale   (basics.go:110-172, :429-470)
//   crypto.CAdd / CSub                          (basics.go:560-590)
//   crypto.InnerSumAll                          (basics.go:278-292)
// ct layout [2][level+1][N], NTT domain, canonical residues.  lattigo restated: the tensor product is the schoolbook
// degree-2 product, relinearisation is the hybrid key switch of rotate.hip with the relinearisation key and the identity
// automorphism, Rescale is ring.DivRoundByLastModulusNTT (floor((x + (q_L-1)/2) / q_L)).
#include "common.hpp"
#include "kernels.hpp"

struct RescaleConst { u64 qL, half; u64 q[SFG_MAXMOD], hneg[SFG_MAXMOD]; double qLinv[SFG_MAXMOD], qLinv_q[SFG_MAXMOD]; };

__device__ __forceinline__ double mm2(double a, double b, double q, double qinv) {     // canonical a*b mod q
    double h = a * b, l = __builtin_fma(a, b, -h);
    double r = __builtin_fma(-__builtin_rint(h * qinv), q, h) + l;
    return canon(r, q, qinv);
}

// grid (N/256, nl, nct): tmp = (a0*b0, a1*b1) as a ciphertext-shaped pair, mid = a0*b1 + a1*b0
__global__ void __launch_bounds__(256) k_tensor(const u64 *a, const u64 *b, u64 *tmp, u64 *mid, int nl, const ModConst *modc) {
    const int N = SFG_N, x = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y; const size_t c = blockIdx.z;
    const double q = modc[m].q, qinv = modc[m].qinv;
    const size_t i0 = ((c * 2 + 0) * nl + m) * N + x, i1 = ((c * 2 + 1) * nl + m) * N + x;
    const double a0 = u64_to_f64(a[i0]), a1 = u64_to_f64(a[i1]), b0 = u64_to_f64(b[i0]), b1 = u64_to_f64(b[i1]);
    tmp[i0] = f64_to_u64(mm2(a0, b0, q, qinv));
    tmp[i1] = f64_to_u64(mm2(a1, b1, q, qinv));
    double s = mm2(a0, b1, q, qinv) + mm2(a1, b0, q, qinv);
    mid[(c * nl + m) * N + x] = f64_to_u64(s >= q ? s - q : s);
}
// grid (N/256, 2*nl, nct)
__global__ void __launch_bounds__(256) k_mul_plain(const u64 *ct, const u64 *pt, size_t pt_stride, u64 *out, int nl, const ModConst *modc) {
    const int N = SFG_N, x = blockIdx.x * 256 + threadIdx.x, row = blockIdx.y, m = row % nl; const size_t c = blockIdx.z;
    const double q = modc[m].q, qinv = modc[m].qinv;
    const size_t i = (c * 2 * nl + row) * N + x;
    out[i] = f64_to_u64(mm2(u64_to_f64(ct[i]), u64_to_f64(pt[c * pt_stride + (size_t)m * N + x]), q, qinv));
}
__global__ void __launch_bounds__(256) k_ct_sub(const u64 *a, const u64 *b, u64 *out, int nl, const ModConst *modc) {
    const int N = SFG_N; const size_t row = blockIdx.x / (N / 256); const int m = (int)(row % nl);
    const u64 q = modc[m].qi;
    const size_t off = row * N + (blockIdx.x % (N / 256)) * 256 + threadIdx.x;
    u64 x = a[off], y = b[off]; out[off] = x >= y ? x - y : x + q - y;
}
// grid (N/256, level, 2*nct): t = INTT(last row) -> ((t + half) mod qL) mod q_m + (q_m - half mod q_m)
__global__ void __launch_bounds__(256) k_rescale_prep(const u64 *t, u64 *tmp, int level, RescaleConst rc) {
    const int N = SFG_N, x = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y; const size_t pb = blockIdx.z;
    u64 v = t[pb * N + x] + rc.half; if (v >= rc.qL) v -= rc.qL;
    u64 w = v % rc.q[m] + rc.hneg[m]; if (w >= rc.q[m]) w -= rc.q[m];
    tmp[(pb * level + m) * N + x] = w;
}
// grid (N/256, level, 2*nct): out = (src - tmp) * qL^-1 mod q_m
__global__ void __launch_bounds__(256) k_rescale_fin(const u64 *ct, const u64 *tmp, u64 *out, int level, RescaleConst rc, const ModConst *modc) {
    const int N = SFG_N, x = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y; const size_t pb = blockIdx.z;
    const double q = modc[m].q, qinv = modc[m].qinv;
    const double d = u64_to_f64(ct[(pb * (level + 1) + m) * N + x]) - u64_to_f64(tmp[(pb * level + m) * N + x]);
    out[(pb * level + m) * N + x] = f64_to_u64(canon(mulmod_lazy(d, rc.qLinv[m], rc.qLinv_q[m], q), q, qinv));
}

struct ScalarRow { u64 c[SFG_MAXMOD]; };
// grid (N/256, 2*nl, nct): out = ct * c[m]  (MultByConst)
__global__ void __launch_bounds__(256) k_mul_scalar(const u64 *ct, ScalarRow sc, u64 *out, int nl, const ModConst *modc) {
    const int N = SFG_N, x = blockIdx.x * 256 + threadIdx.x, row = blockIdx.y, m = row % nl; const size_t c = blockIdx.z;
    const double q = modc[m].q, qinv = modc[m].qinv;
    const size_t i = (c * 2 * nl + row) * N + x;
    out[i] = f64_to_u64(mm2(u64_to_f64(ct[i]), u64_to_f64(sc.c[m]), q, qinv));
}
// grid (N/256, 2*nl, nct): acc += ct * c[m]  (MultByConstAndAdd's arithmetic: ring.MRed(p0, MForm(c)) + CRed)
__global__ void __launch_bounds__(256) k_mul_scalar_add(const u64 *ct, ScalarRow sc, u64 *acc, int nl, const ModConst *modc) {
    const int N = SFG_N, x = blockIdx.x * 256 + threadIdx.x, row = blockIdx.y, m = row % nl; const size_t c = blockIdx.z;
    const double q = modc[m].q, qinv = modc[m].qinv;
    const size_t i = (c * 2 * nl + row) * N + x;
    double v = mm2(u64_to_f64(ct[i]), u64_to_f64(sc.c[m]), q, qinv) + u64_to_f64(acc[i]);
    acc[i] = f64_to_u64(v >= q ? v - q : v);
}
// grid (N/256, 2*nl, nct): out = ct, with c[m] (AddConst) or pt[m][x] (AddNew(ct, plaintext)) added to polynomial 0
__global__ void __launch_bounds__(256) k_add_c0(const u64 *ct, ScalarRow sc, const u64 *pt, size_t pt_stride, u64 *out, int nl, const ModConst *modc) {
    const int N = SFG_N, x = blockIdx.x * 256 + threadIdx.x, row = blockIdx.y, m = row % nl; const size_t c = blockIdx.z;
    const u64 q = modc[m].qi;
    const size_t i = (c * 2 * nl + row) * N + x;
    u64 v = ct[i];
    if (row < nl) { v += pt ? pt[c * pt_stride + (size_t)m * N + x] : sc.c[m]; if (v >= q) v -= q; }
    out[i] = v;
}

static int check_level(sfg_ctx *ctx, int level, int nct) {
    if (level < 0 || level >= ctx->nq) SFG_FAIL(ctx, "evaluator op: level %d out of range", level);
    if (nct < 0) SFG_FAIL(ctx, "evaluator op: negative ciphertext count");
    return 0;
}

extern "C" int sfg_ctx_load_relinkey(sfg_ctx *ctx, const uint64_t *key_host, int mont) {
    // the relinearisation key is a switching key used with the identity automorphism: stored under Galois element 1
    return sfg_ctx_load_rotkey(ctx, 1, key_host, mont);
}

extern "C" int sfg_ct_sub_dev(sfg_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, int nct, int level) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    SFG_TRY(check_level(ctx, level, nct));
    const int nl = level + 1; const size_t rows = (size_t)nct * 2 * nl;
    if (!rows) return 0;
    hipLaunchKernelGGL(k_ct_sub, dim3((unsigned)(rows * (SFG_N / 256))), dim3(256), 0, ctx->stream, (const u64 *)a, (const u64 *)b, (u64 *)out, nl, ctx->modc);
    SFG_HIP(ctx, hipGetLastError());
    return 0;
}

extern "C" int sfg_ct_mulrelin_dev(sfg_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, int nct, int level) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    SFG_TRY(check_level(ctx, level, nct));
    if (!nct) return 0;
    if (!sfg_ctx_has_rotkey(ctx, 1)) SFG_FAIL(ctx, "mulrelin: no relinearisation key loaded (sfg_ctx_load_relinkey)");
    const int N = SFG_N, nl = level + 1;
    PhaseTimer t(ctx, "mulrelin");
    void *p;
    SFG_TRY(sfg_scratch(ctx, "ev_tensor", (size_t)nct * 3 * nl * N * 8, &p));
    u64 *tmp = (u64 *)p, *mid = tmp + (size_t)nct * 2 * nl * N;
    hipLaunchKernelGGL(k_tensor, dim3(N / 256, nl, nct), dim3(256), 0, ctx->stream, (const u64 *)a, (const u64 *)b, tmp, mid, nl, ctx->modc);
    SFG_HIP(ctx, hipGetLastError());
    int rc = launch_relinearize(ctx, tmp, nct, level, mid, (u64 *)out);
    t.stop(1);
    return rc;
}

extern "C" int sfg_ct_mul_plain_dev(sfg_ctx *ctx, const uint64_t *ct, const uint64_t *pt, size_t pt_stride, uint64_t *out, int nct, int level) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    SFG_TRY(check_level(ctx, level, nct));
    if (!nct) return 0;
    const int N = SFG_N, nl = level + 1;
    hipLaunchKernelGGL(k_mul_plain, dim3(N / 256, 2 * nl, nct), dim3(256), 0, ctx->stream, (const u64 *)ct, (const u64 *)pt, pt_stride, (u64 *)out, nl, ctx->modc);
    SFG_HIP(ctx, hipGetLastError());
    return 0;
}

extern "C" int sfg_ct_rescale_dev(sfg_ctx *ctx, const uint64_t *in, uint64_t *out, int nct, int level) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    SFG_TRY(check_level(ctx, level, nct));
    if (level == 0) SFG_FAIL(ctx, "rescale: input ciphertext already at level 0");        // lattigo's error, evaluator.Rescale
    if (!nct) return 0;
    const int N = SFG_N, nl = level + 1; const size_t np2 = (size_t)nct * 2;
    RescaleConst rc; memset(&rc, 0, sizeof rc);
    rc.qL = ctx->q[level]; rc.half = (rc.qL - 1) >> 1;
    for (int m = 0; m < level; m++) {
        u64 q = ctx->q[m]; rc.q[m] = q; rc.hneg[m] = q - rc.half % q;
        u64 inv = h_invmod(rc.qL % q, q); rc.qLinv[m] = (double)inv; rc.qLinv_q[m] = (double)inv / (double)q;
    }
    PhaseTimer t(ctx, "rescale");
    void *p;
    SFG_TRY(sfg_scratch(ctx, "ev_rescale", np2 * (1 + (size_t)level) * N * 8, &p));
    u64 *tl = (u64 *)p, *tmp = tl + np2 * N;
    ModPattern pl; pl.period = 1; pl.m[0] = (int8_t)level;
    RowMap rm; rm.rpg = 1; rm.gstride_in = (size_t)nl * N; rm.gstride_out = N;
    SFG_TRY(launch_ntt_inv_map(ctx, (const u64 *)in + (size_t)level * N, tl, np2, pl, rm));
    hipLaunchKernelGGL(k_rescale_prep, dim3(N / 256, level, (unsigned)np2), dim3(256), 0, ctx->stream, tl, tmp, level, rc);
    SFG_HIP(ctx, hipGetLastError());
    ModPattern pq; pq.period = level; for (int m = 0; m < level; m++) pq.m[m] = (int8_t)m;
    SFG_TRY(launch_ntt_fwd(ctx, tmp, tmp, np2 * level, pq));
    hipLaunchKernelGGL(k_rescale_fin, dim3(N / 256, level, (unsigned)np2), dim3(256), 0, ctx->stream, (const u64 *)in, tmp, (u64 *)out, level, rc, ctx->modc);
    SFG_HIP(ctx, hipGetLastError());
    t.stop(1);
    return 0;
}

// InnerSumAll: out = sum over the nct inputs and over all slots (every slot of `out` holds the total)
extern "C" int sfg_ct_innersum_dev(sfg_ctx *ctx, const uint64_t *in, int nct, int level, uint64_t *out) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    SFG_TRY(check_level(ctx, level, nct));
    if (nct < 1) SFG_FAIL(ctx, "innersum: needs at least one ciphertext");
    const int N = SFG_N, nl = level + 1; const size_t ctw = (size_t)2 * nl * N;
    PhaseTimer t(ctx, "innersum");
    void *p;
    SFG_TRY(sfg_scratch(ctx, "ev_innersum", ctw * 8, &p));
    u64 *rt = (u64 *)p, *o = (u64 *)out;
    SFG_HIP(ctx, hipMemcpyAsync(o, in, ctw * 8, hipMemcpyDeviceToDevice, ctx->stream));
    for (int i = 1; i < nct; i++) SFG_TRY(launch_ct_add(ctx, (const u64 *)in + (size_t)i * ctw, o, o, 1, level));
    for (int rot = 1; rot < SFG_SLOTS; rot *= 2) {                        // basics.go:283-289: RotateAndAdd by 1,2,4,.. (left)
        int nrot = SFG_SLOTS - rot;                                         // left by rot == right by slots - rot
        SFG_TRY(launch_rotate_right(ctx, o, rt, 1, level, &nrot));
        SFG_TRY(launch_ct_add(ctx, rt, o, o, 1, level));
    }
    t.stop(1);
    return 0;
}

// eval.MultByConst: both polynomials times one residue per modulus (scalars_host[level+1], canonical; see crypto::CMultConst in
// the host mirror for lattigo's scaleUpExact rule that produces them and the scale bookkeeping)
extern "C" int sfg_ct_mul_scalar_dev(sfg_ctx *ctx, const uint64_t *ct, const uint64_t *scalars_host, uint64_t *out, int nct, int level) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    SFG_TRY(check_level(ctx, level, nct));
    if (!nct) return 0;
    const int N = SFG_N, nl = level + 1;
    ScalarRow sc; memset(&sc, 0, sizeof sc);
    for (int m = 0; m < nl; m++) { if (scalars_host[m] >= ctx->q[m]) SFG_FAIL(ctx, "mul_scalar: residue %d not canonical", m); sc.c[m] = scalars_host[m]; }
    hipLaunchKernelGGL(k_mul_scalar, dim3(N / 256, 2 * nl, nct), dim3(256), 0, ctx->stream, (const u64 *)ct, sc, (u64 *)out, nl, ctx->modc);
    SFG_HIP(ctx, hipGetLastError());
    return 0;
}
// eval.MultByConstAndAdd's arithmetic (pca.go:264, qrfact.go:195,280): acc += ct * scalars[m] on both polynomials; the scale matching that
// precedes it in lattigo is host-side bookkeeping (crypto::MultByConstAndAddDev in the host mirror)
extern "C" int sfg_ct_mul_scalar_add_dev(sfg_ctx *ctx, const uint64_t *ct, const uint64_t *scalars_host, uint64_t *acc, int nct, int level) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    SFG_TRY(check_level(ctx, level, nct));
    if (!nct) return 0;
    const int N = SFG_N, nl = level + 1;
    ScalarRow sc; memset(&sc, 0, sizeof sc);
    for (int m = 0; m < nl; m++) { if (scalars_host[m] >= ctx->q[m]) SFG_FAIL(ctx, "mul_scalar_add: residue %d not canonical", m); sc.c[m] = scalars_host[m]; }
    hipLaunchKernelGGL(k_mul_scalar_add, dim3(N / 256, 2 * nl, nct), dim3(256), 0, ctx->stream, (const u64 *)ct, sc, (u64 *)acc, nl, ctx->modc);
    SFG_HIP(ctx, hipGetLastError());
    return 0;
}
// eval.AddConst: one residue per modulus added to every NTT coefficient of polynomial 0
extern "C" int sfg_ct_add_scalar_dev(sfg_ctx *ctx, const uint64_t *ct, const uint64_t *scalars_host, uint64_t *out, int nct, int level) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    SFG_TRY(check_level(ctx, level, nct));
    if (!nct) return 0;
    const int N = SFG_N, nl = level + 1;
    ScalarRow sc; memset(&sc, 0, sizeof sc);
    for (int m = 0; m < nl; m++) { if (scalars_host[m] >= ctx->q[m]) SFG_FAIL(ctx, "add_scalar: residue %d not canonical", m); sc.c[m] = scalars_host[m]; }
    hipLaunchKernelGGL(k_add_c0, dim3(N / 256, 2 * nl, nct), dim3(256), 0, ctx->stream, (const u64 *)ct, sc, (const u64 *)nullptr, (size_t)0, (u64 *)out, nl, ctx->modc);
    SFG_HIP(ctx, hipGetLastError());
    return 0;
}
// eval.AddNew(ct, plaintext): polynomial 0 += pt (NTT domain, [level+1][N]; pt_stride words between plaintexts, 0 = shared)
extern "C" int sfg_ct_add_plain_dev(sfg_ctx *ctx, const uint64_t *ct, const uint64_t *pt, size_t pt_stride, uint64_t *out, int nct, int level) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    SFG_TRY(check_level(ctx, level, nct));
    if (!nct) return 0;
    const int N = SFG_N, nl = level + 1;
    ScalarRow sc; memset(&sc, 0, sizeof sc);
    hipLaunchKernelGGL(k_add_c0, dim3(N / 256, 2 * nl, nct), dim3(256), 0, ctx->stream, (const u64 *)ct, sc, (const u64 *)pt, pt_stride, (u64 *)out, nl, ctx->modc);
    SFG_HIP(ctx, hipGetLastError());
    return 0;
}
