// refresh.hip — local work of the collective bootstrap: mpc/mhe.go:222-348 brackets every hot matrix product with
// CollectiveBootstrap(Mat), which per ciphertext calls lattigo's dckks.RefreshProtocol: GenShares (mhe.go:251,315), then - after the network
// aggregation of the shares - Decrypt, Recode, Recrypt (mhe.go:256-258,329-331).  With the products at ~18 s per power iteration this is the next
// local cost, and it is the reason ciphertexts would otherwise leave the device between the two products.
//
// PARITY UNPINNED: RefreshProtocol lives in the absent fork (github.com/hcholab/lattigo/v2, go.mod:5,12).  Restated from the published
// lattigo v2.1.0 dckks/refresh.go (the CPU checker under oracle/ states it function by function) for the case "ciphertext scale == target scale"; the fork's extra
// target-scale argument (mhe.go:315,330) is not in the reference tree.  Randomness (mask, e0, e1, crs) is an input: the Go side keeps drawing it.
//
//   GenShares:  h0 = NTT_level(mask + e0) + sk (.) c1              h1 = -( NTT(mask + e1) + sk (.) crs )        (NTT is linear: one transform per row)
//   finish:     x  = INTT_level(c0 + sum h0);  big integer by CRT, recentred at Q_level / 2 (PolyToBigint + the Cmp(QHalf) rule), reduced into
//               all nq moduli (here: Garner mixed-radix digits, digit-wise comparison, Horner per new modulus - no multi-word arithmetic),
//               c0' = NTT(x) + sum h1,  c1' = crs.
#include "common.hpp"
#include "kernels.hpp"

constexpr int RF_MAXL = 12;       // moduli of the input level
struct RecodeConst {
    int nl, nq;
    double inv[RF_MAXL][RF_MAXL];         // inv[i][t] = q_t^-1 mod q_i, t < i
    double half[RF_MAXL];                 // mixed-radix digits of floor(Q_level / 2)
    double qm[SFG_MAXMOD][RF_MAXL];       // q_i mod q_j for the new moduli j >= nl
    double Qmod[SFG_MAXMOD];              // Q_level mod q_j
};

// rows[(ct, j)][x] = (mask + e)[ct][x] mod q_j, coefficient domain.  mask: [nct][N][W] two's-complement 64-bit limbs; e: [nct][N] int32.
// Horner over 32-bit digits in exact fp64: acc * 2^32 + digit.  grid (N/256, nmod, nct)
__global__ void __launch_bounds__(256) k_bigint_rows(const u64 *mask, int W, const int *e, u64 *rows, int nmod, const ModConst *modc) {
    const int N = SFG_N, x = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y; const size_t c = blockIdx.z;
    const double q = modc[j].q, qinv = modc[j].qinv;
    const double B = canon(4294967296.0, q, qinv), Bq = B * qinv;
    const u64 *src = mask + (c * N + x) * (size_t)W;
    const bool neg = (src[W - 1] >> 63) != 0;
    double acc = 0.0;                                    // magnitude of a negative value = ~limbs + 1: evaluate ~limbs top-down, add the 1 modulo q at the end
    for (int i = W - 1; i >= 0; i--) {
        const u64 w = neg ? ~src[i] : src[i];
        acc = canon(mulmod_lazy(acc, B, Bq, q) + (double)(unsigned)(w >> 32), q, qinv);
        acc = canon(mulmod_lazy(acc, B, Bq, q) + (double)(unsigned)w, q, qinv);
    }
    if (neg) { acc += 1.0; acc = acc >= q ? acc - q : acc; acc = acc == 0.0 ? 0.0 : q - acc; }
    double v = acc + (double)e[c * N + x];              // |e| << q
    v = v < 0.0 ? v + q : v; v = v >= q ? v - q : v;
    rows[(c * nmod + j) * (size_t)N + x] = f64_to_u64(v);
}
// rows[(ct, j)][x] = e[ct][x] mod q_j (small signed coefficients).  grid (N/256, nmod, nct)
__global__ void __launch_bounds__(256) k_small_rows(const int *e, u64 *rows, int nmod, const ModConst *modc) {
    const int N = SFG_N, x = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y; const size_t c = blockIdx.z;
    const long long v = e[c * N + x];
    rows[(c * nmod + j) * (size_t)N + x] = v < 0 ? modc[j].qi - (u64)(-v) : (u64)v;
}
// h = rows + sk (.) xrow (mod q), negated if neg.  rows/h: [nct][nmod][N]; xrow: row (ct, j) at x + ct * x_ct_stride + j * N.  grid (N/256, nmod, nct)
__global__ void __launch_bounds__(256) k_share(const u64 *rows, const u64 *sk, const u64 *xr, size_t x_ct_stride, u64 *h, int nmod, int neg, const ModConst *modc) {
    const int N = SFG_N, x = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y; const size_t c = blockIdx.z;
    const double q = modc[j].q, qinv = modc[j].qinv;
    const size_t i = (c * nmod + j) * (size_t)N + x;
    const double s = u64_to_f64(sk[(size_t)j * N + x]), v = u64_to_f64(xr[c * x_ct_stride + (size_t)j * N + x]);
    const double hh = s * v, ll = __builtin_fma(s, v, -hh);
    double r = canon(__builtin_fma(-__builtin_rint(hh * qinv), q, hh) + ll, q, qinv) + u64_to_f64(rows[i]);
    r = r >= q ? r - q : r;
    if (neg) r = r == 0.0 ? 0.0 : q - r;
    h[i] = f64_to_u64(r);
}
// out[(ct, j)] = a[(ct, j)] + b[(ct, j)] mod q_j over [nct][nl][N] rows taken from strided sources.  grid (N/256, nl, nct)
__global__ void __launch_bounds__(256) k_add_rows(const u64 *a, size_t a_ct_stride, const u64 *b, size_t b_ct_stride, u64 *out, size_t out_ct_stride, const ModConst *modc) {
    const int N = SFG_N, x = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y; const size_t c = blockIdx.z;
    const u64 q = modc[j].qi;
    u64 v = a[c * a_ct_stride + (size_t)j * N + x] + b[c * b_ct_stride + (size_t)j * N + x];
    out[c * out_ct_stride + (size_t)j * N + x] = v >= q ? v - q : v;
}
// Recode of one coefficient per thread.  xin: [nct][nl][N] coefficient-domain residues of x in [0, Q); out: polynomial 0 of [nct][2][nq][N], coefficient domain.
// grid (N/256, nct)
__global__ void __launch_bounds__(256) k_recode(const u64 *xin, u64 *out, RecodeConst rc, const ModConst *modc) {
    const int N = SFG_N, x = blockIdx.x * 256 + threadIdx.x; const size_t c = blockIdx.y;
    const int nl = rc.nl, nq = rc.nq;
    double v[RF_MAXL], r[RF_MAXL];
    for (int i = 0; i < nl; i++) r[i] = u64_to_f64(xin[(c * nl + i) * (size_t)N + x]);
    // Garner: x = v0 + v1 q0 + v2 q0 q1 + ..., 0 <= v_i < q_i
    for (int i = 0; i < nl; i++) {
        const double q = modc[i].q, qinv = modc[i].qinv;
        double t = r[i];
        for (int s = 0; s < i; s++) {
            const double d = t - canon(v[s], q, qinv);                         // (-q, q)
            t = canon(mulmod_lazy(d, rc.inv[i][s], rc.inv[i][s] * qinv, q), q, qinv);
        }
        v[i] = t;
    }
    // x >= floor(Q/2)  (lattigo: Cmp(QHalf) is 1 or 0)  ->  the represented value is x - Q
    bool neg = true;                                                           // all digits equal: x == QHalf counts as negative
    for (int i = nl - 1; i >= 0; i--) if (v[i] != rc.half[i]) { neg = v[i] > rc.half[i]; break; }
    u64 *o = out + c * 2 * nq * (size_t)N + x;
    for (int j = 0; j < nl; j++) o[(size_t)j * N] = f64_to_u64(r[j]);           // x and x - Q agree modulo the moduli of Q
    for (int j = nl; j < nq; j++) {
        const double q = modc[j].q, qinv = modc[j].qinv;
        double acc = canon(v[nl - 1], q, qinv);
        for (int i = nl - 2; i >= 0; i--) acc = canon(mulmod_lazy(acc, rc.qm[j][i], rc.qm[j][i] * qinv, q) + canon(v[i], q, qinv), q, qinv);
        if (neg) { acc -= rc.Qmod[j]; acc = acc < 0.0 ? acc + q : acc; }
        o[(size_t)j * N] = f64_to_u64(acc);
    }
}
// Recrypt: c0 += h1agg, c1 = crs.  out [nct][2][nq][N]; h1agg, crs [nct][nq][N].  grid (N/256, nq, nct)
__global__ void __launch_bounds__(256) k_recrypt(u64 *out, const u64 *h1, const u64 *crs, int nq, const ModConst *modc) {
    const int N = SFG_N, x = blockIdx.x * 256 + threadIdx.x, j = blockIdx.y; const size_t c = blockIdx.z;
    const u64 q = modc[j].qi;
    const size_t s = (c * nq + j) * (size_t)N + x, o = (c * 2 * nq + j) * (size_t)N + x;
    u64 v = out[o] + h1[s];
    out[o] = v >= q ? v - q : v;
    out[o + (size_t)nq * N] = crs[s];
}

// ---- target-scale form (what the reference calls: mhe.go:251,256-258,315,329-331 pass parameters.Scale() while the products they refresh carry
// A.scale * Delta, matmult.go:1045).  lattigo v2.2.0 dckks/refresh.go restated (PARITY UNPINNED; the CPU checker under oracle/ states it function by function):
//   GenShares: the recrypt share is built from Quo(mask * Int(target), Int(ct scale));  Recode: x <- Quo(x * Int(target), Int(ct scale)) before the
//   re-reduction into all nq moduli.  Quo truncates towards zero: floor on the magnitude, sign kept.  Int(float64) = m * 2^e exactly (m < 2^53), so the
//   ratio is one multiplication by m_out, one shift by e_out - e_in and one short division by m_in on a 512-bit magnitude, one coefficient per lane.
constexpr int BG = 8;                                    // 64-bit limbs of a device big integer
struct ScaleRatio { u64 mo, mi; int sh; };               // |x| <- floor(|x| * mo * 2^sh / mi)
__device__ __forceinline__ void bg_mul_add(u64 (&a)[BG], u64 m, u64 add) {          // a = a * m + add
    u64 c = add;
#pragma unroll
    for (int i = 0; i < BG; i++) { const u64 lo = a[i] * m, hi = __umul64hi(a[i], m); const u64 s = lo + c; a[i] = s; c = hi + (s < lo); }
}
__device__ __forceinline__ void bg_rsub(u64 (&a)[BG], const u64 *b) {               // a = b - a   (b >= a)
    u64 br = 0;
#pragma unroll
    for (int i = 0; i < BG; i++) { const u64 bi = b[i], d = bi - a[i], d2 = d - br; br = (bi < a[i]) | (d < br); a[i] = d2; }
}
__device__ __forceinline__ void bg_shift(u64 (&a)[BG], int sh) {                    // sh > 0: left, sh < 0: right (uniform per launch)
    if (!sh) return;
    u64 t[BG];
    const int k = sh > 0 ? sh : -sh, ws = k >> 6, bs = k & 63;
#pragma unroll
    for (int i = 0; i < BG; i++) {
        u64 v = 0;
        if (sh > 0) { const int s0 = i - ws; const u64 hi = s0 >= 0 ? a[s0] : 0, lo = s0 - 1 >= 0 ? a[s0 - 1] : 0; v = bs ? (hi << bs) | (lo >> (64 - bs)) : hi; }
        else { const int s0 = i + ws; const u64 lo = s0 < BG ? a[s0] : 0, hi = s0 + 1 < BG ? a[s0 + 1] : 0; v = bs ? (lo >> bs) | (hi << (64 - bs)) : lo; }
        t[i] = v;
    }
#pragma unroll
    for (int i = 0; i < BG; i++) a[i] = t[i];
}
__device__ __forceinline__ void bg_div_small(u64 (&a)[BG], u64 d) {                 // a = floor(a / d), d < 2^53: a byte at a time, r * 256 + byte < 2^61
    if (d == 1) return;
    u64 r = 0;
#pragma unroll
    for (int i = BG - 1; i >= 0; i--) {
        u64 w = a[i], q = 0;
#pragma unroll
        for (int b = 7; b >= 0; b--) { r = (r << 8) | ((w >> (8 * b)) & 0xFF); const u64 qb = r / d; r -= qb * d; q = (q << 8) | qb; }
        a[i] = q;
    }
}
__device__ __forceinline__ void bg_rescale(u64 (&a)[BG], const ScaleRatio &sr) { bg_mul_add(a, sr.mo, 0); bg_shift(a, sr.sh); bg_div_small(a, sr.mi); }
__device__ __forceinline__ double bg_mod(const u64 (&a)[BG], double q, double qinv) {   // Horner over 32-bit digits in exact fp64
    const double B = canon(4294967296.0, q, qinv), Bq = B * qinv;
    double acc = 0.0;
#pragma unroll
    for (int i = BG - 1; i >= 0; i--) {
        acc = canon(mulmod_lazy(acc, B, Bq, q) + (double)(unsigned)(a[i] >> 32), q, qinv);
        acc = canon(mulmod_lazy(acc, B, Bq, q) + (double)(unsigned)a[i], q, qinv);
    }
    return acc;
}
// rows[(ct, j)][x] = (Quo(mask * out, in) + e)[ct][x] mod q_j for all nmod moduli.  grid (N/256, nct)
__global__ void __launch_bounds__(256) k_bigint_rows_scaled(const u64 *mask, int W, const int *e, u64 *rows, int nmod, ScaleRatio sr, const ModConst *modc) {
    const int N = SFG_N, x = blockIdx.x * 256 + threadIdx.x; const size_t c = blockIdx.y;
    const u64 *src = mask + (c * N + x) * (size_t)W;
    const bool neg = (src[W - 1] >> 63) != 0;
    u64 a[BG];
#pragma unroll
    for (int i = 0; i < BG; i++) a[i] = i < W ? (neg ? ~src[i] : src[i]) : 0;
    if (neg) { u64 cy = 1; for (int i = 0; i < BG && cy; i++) { a[i] += cy; cy = a[i] == 0; } for (int i = W; i < BG; i++) a[i] = 0; }
    bg_rescale(a, sr);
    const double ev = (double)e[c * N + x];
    for (int j = 0; j < nmod; j++) {
        const double q = modc[j].q, qinv = modc[j].qinv;
        double v = bg_mod(a, q, qinv);
        if (neg) v = v == 0.0 ? 0.0 : q - v;
        v += ev; v = v < 0.0 ? v + q : v; v = v >= q ? v - q : v;
        rows[(c * nmod + j) * (size_t)N + x] = f64_to_u64(v);
    }
}
struct RecodeBig { u64 Q[BG]; u64 qi[RF_MAXL]; };
// Recode with the scale ratio: every one of the nq output rows comes from the rescaled big integer.  grid (N/256, nct)
__global__ void __launch_bounds__(256) k_recode_scaled(const u64 *xin, u64 *out, RecodeConst rc, RecodeBig rb, ScaleRatio sr, const ModConst *modc) {
    const int N = SFG_N, x = blockIdx.x * 256 + threadIdx.x; const size_t c = blockIdx.y;
    const int nl = rc.nl, nq = rc.nq;
    double v[RF_MAXL];
    for (int i = 0; i < nl; i++) {
        const double q = modc[i].q, qinv = modc[i].qinv;
        double t = u64_to_f64(xin[(c * nl + i) * (size_t)N + x]);
        for (int s = 0; s < i; s++) {
            const double d = t - canon(v[s], q, qinv);
            t = canon(mulmod_lazy(d, rc.inv[i][s], rc.inv[i][s] * qinv, q), q, qinv);
        }
        v[i] = t;
    }
    bool neg = true;
    for (int i = nl - 1; i >= 0; i--) if (v[i] != rc.half[i]) { neg = v[i] > rc.half[i]; break; }
    u64 a[BG];
#pragma unroll
    for (int i = 0; i < BG; i++) a[i] = 0;
    a[0] = f64_to_u64(v[nl - 1]);
    for (int i = nl - 2; i >= 0; i--) bg_mul_add(a, rb.qi[i], f64_to_u64(v[i]));      // x = v0 + q0 (v1 + q1 (v2 + ...))
    if (neg) bg_rsub(a, rb.Q);                                                         // |x - Q|
    bg_rescale(a, sr);
    u64 *o = out + c * 2 * nq * (size_t)N + x;
    for (int j = 0; j < nq; j++) {
        const double q = modc[j].q, qinv = modc[j].qinv;
        double r = bg_mod(a, q, qinv);
        if (neg) r = r == 0.0 ? 0.0 : q - r;
        o[(size_t)j * N] = f64_to_u64(r);
    }
}
// Int(big.Float(f)) = m * 2^e, m < 2^53, for a finite scale f >= 1
static int scale_int(sfg_ctx *ctx, double f, u64 &m, int &e) {
    if (!(f >= 1.0) || f > 0x1p400) SFG_FAIL(ctx, "refresh: scale %g out of range", f);
    int ex; const double fr = frexp(f, &ex);
    m = (u64)ldexp(fr, 53); e = ex - 53;
    if (e < 0) { m >>= -e; e = 0; }
    return 0;
}
static int scale_ratio(sfg_ctx *ctx, int level, double ct_scale, double target_scale, ScaleRatio &sr) {
    u64 mo, mi; int eo, ei;
    SFG_TRY(scale_int(ctx, target_scale, mo, eo)); SFG_TRY(scale_int(ctx, ct_scale, mi, ei));
    double bits = 0; for (int i = 0; i <= level; i++) bits += log2((double)ctx->q[i]);
    if (bits + 54 + (eo > ei ? eo - ei : 0) > 64.0 * BG - 2) SFG_FAIL(ctx, "refresh: Q_level * target scale / ciphertext scale does not fit %d bits", 64 * BG);
    sr.mo = mo; sr.mi = mi; sr.sh = eo - ei;
    return 0;
}

static int refresh_check(sfg_ctx *ctx, int nct, int level) {
    if (level < 0 || level >= ctx->nq) SFG_FAIL(ctx, "refresh: level %d out of range", level);
    if (level + 1 > RF_MAXL) SFG_FAIL(ctx, "refresh: more than %d moduli at the input level", RF_MAXL);
    if (nct < 0) SFG_FAIL(ctx, "refresh: negative ciphertext count");
    return 0;
}

static int refresh_gen_shares(sfg_ctx *ctx, const uint64_t *ct, int nct, int level, const uint64_t *crs, const uint64_t *mask, int W,
                              const int32_t *e0, const int32_t *e1, uint64_t *h0, uint64_t *h1, const ScaleRatio *sr);
extern "C" int sfg_refresh_gen_shares_dev(sfg_ctx *ctx, const uint64_t *ct, int nct, int level, const uint64_t *crs, const uint64_t *mask, int W,
                                          const int32_t *e0, const int32_t *e1, uint64_t *h0, uint64_t *h1) {
    return refresh_gen_shares(ctx, ct, nct, level, crs, mask, W, e0, e1, h0, h1, nullptr);
}
extern "C" int sfg_refresh_gen_shares_scaled_dev(sfg_ctx *ctx, const uint64_t *ct, int nct, int level, double ct_scale, double target_scale, const uint64_t *crs,
                                                 const uint64_t *mask, int W, const int32_t *e0, const int32_t *e1, uint64_t *h0, uint64_t *h1) {
    if (level < 0 || level >= ctx->nq) SFG_FAIL(ctx, "refresh: level %d out of range", level);
    if (ct_scale == target_scale) return refresh_gen_shares(ctx, ct, nct, level, crs, mask, W, e0, e1, h0, h1, nullptr);     // ratio 1: the unscaled kernels (Quo(mask * x, x) = mask)
    ScaleRatio sr; SFG_TRY(scale_ratio(ctx, level, ct_scale, target_scale, sr));
    if (W > BG) SFG_FAIL(ctx, "refresh: mask limb count %d exceeds %d in the target-scale form", W, BG);
    // the kernel multiplies the W-limb mask magnitude by the 53-bit mantissa of the target scale and shifts it left by (target exponent - ciphertext exponent)
    // inside 64 BG bits: a mask wider than Q (the bound scale_ratio checks) must fit as well, or the product would wrap silently
    if (64 * W + 54 + (sr.sh > 0 ? sr.sh : 0) > 64 * BG) SFG_FAIL(ctx, "refresh: a %d-limb mask times the target scale (shift %d) does not fit %d bits", W, sr.sh, 64 * BG);
    return refresh_gen_shares(ctx, ct, nct, level, crs, mask, W, e0, e1, h0, h1, &sr);
}
static int refresh_gen_shares(sfg_ctx *ctx, const uint64_t *ct, int nct, int level, const uint64_t *crs, const uint64_t *mask, int W,
                              const int32_t *e0, const int32_t *e1, uint64_t *h0, uint64_t *h1, const ScaleRatio *sr) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    SFG_TRY(refresh_check(ctx, nct, level));
    if (!ctx->sh->sk_dev) SFG_FAIL(ctx, "refresh: no secret-key shard loaded (sfg_ctx_load_secret_key)");
    if (W < 1 || W > 16) SFG_FAIL(ctx, "refresh: mask limb count %d out of range", W);
    if (!nct) return 0;
    const int N = SFG_N, nl = level + 1, nq = ctx->nq;
    const u64 *sk = ctx->sh->sk_dev;
    // h0: rows (mask + e0) mod q_j, j <= level -> NTT -> + sk (.) c1
    hipLaunchKernelGGL(k_bigint_rows, dim3(N / 256, nl, nct), dim3(256), 0, ctx->stream, (const u64 *)mask, W, (const int *)e0, (u64 *)h0, nl, ctx->modc);
    SFG_HIP(ctx, hipGetLastError());
    ModPattern p0; p0.period = nl; for (int j = 0; j < nl; j++) p0.m[j] = (int8_t)j;
    SFG_TRY(launch_ntt_fwd(ctx, (const u64 *)h0, (u64 *)h0, (size_t)nct * nl, p0));
    hipLaunchKernelGGL(k_share, dim3(N / 256, nl, nct), dim3(256), 0, ctx->stream, (const u64 *)h0, sk, (const u64 *)ct + (size_t)nl * N, (size_t)2 * nl * N, (u64 *)h0, nl, 0, ctx->modc);
    // h1: all nq moduli, against the common reference polynomial, negated
    if (sr) hipLaunchKernelGGL(k_bigint_rows_scaled, dim3(N / 256, nct), dim3(256), 0, ctx->stream, (const u64 *)mask, W, (const int *)e1, (u64 *)h1, nq, *sr, ctx->modc);
    else hipLaunchKernelGGL(k_bigint_rows, dim3(N / 256, nq, nct), dim3(256), 0, ctx->stream, (const u64 *)mask, W, (const int *)e1, (u64 *)h1, nq, ctx->modc);
    SFG_HIP(ctx, hipGetLastError());
    ModPattern p1; p1.period = nq; for (int j = 0; j < nq; j++) p1.m[j] = (int8_t)j;
    SFG_TRY(launch_ntt_fwd(ctx, (const u64 *)h1, (u64 *)h1, (size_t)nct * nq, p1));
    hipLaunchKernelGGL(k_share, dim3(N / 256, nq, nct), dim3(256), 0, ctx->stream, (const u64 *)h1, sk, (const u64 *)crs, (size_t)nq * N, (u64 *)h1, nq, 1, ctx->modc);
    SFG_HIP(ctx, hipGetLastError());
    return 0;
}

// ---- f-4 (partial): the ring work of MPC.CMatToSS (mpc/ss.go:146-281).  Its mask share is GenShares' h0 (ss.go:222-236: SetCoefficientsBigintLvl,
// NTTLvl, MulCoeffsMontgomeryAndAddLvl(sk, c1), + NTT(gaussian)), and it also keeps NTT(mask) itself as the plaintext ctMask (ss.go:226) that
// DecodeRVec turns into the party's additive share.  DecodeRVec / EncodeRVecNew are fork-only encoder entry points and stay in Go.
//   h0 [nct][level+1][N] = NTT(mask) + sk (.) c1 + NTT(e0);   mask_ntt [nct][level+1][N] = NTT(mask)
extern "C" int sfg_ckks_to_ss_share_dev(sfg_ctx *ctx, const uint64_t *ct, int nct, int level, const uint64_t *mask, int W, const int32_t *e0,
                                        uint64_t *h0, uint64_t *mask_ntt) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    SFG_TRY(refresh_check(ctx, nct, level));
    if (!ctx->sh->sk_dev) SFG_FAIL(ctx, "CMatToSS: no secret-key shard loaded (sfg_ctx_load_secret_key)");
    if (W < 1 || W > 16) SFG_FAIL(ctx, "CMatToSS: mask limb count %d out of range", W);
    if (!nct) return 0;
    const int N = SFG_N, nl = level + 1;
    int *zero = nullptr;
    SFG_TRY(sfg_scratch(ctx, "refresh.zero_e", (size_t)nct * N * sizeof(int), (void **)&zero));
    SFG_HIP(ctx, hipMemsetAsync(zero, 0, (size_t)nct * N * sizeof(int), ctx->stream));
    ModPattern p0; p0.period = nl; for (int j = 0; j < nl; j++) p0.m[j] = (int8_t)j;
    hipLaunchKernelGGL(k_bigint_rows, dim3(N / 256, nl, nct), dim3(256), 0, ctx->stream, (const u64 *)mask, W, (const int *)zero, (u64 *)mask_ntt, nl, ctx->modc);
    SFG_HIP(ctx, hipGetLastError());
    SFG_TRY(launch_ntt_fwd(ctx, (const u64 *)mask_ntt, (u64 *)mask_ntt, (size_t)nct * nl, p0));
    hipLaunchKernelGGL(k_small_rows, dim3(N / 256, nl, nct), dim3(256), 0, ctx->stream, (const int *)e0, (u64 *)h0, nl, ctx->modc);
    SFG_HIP(ctx, hipGetLastError());
    SFG_TRY(launch_ntt_fwd(ctx, (const u64 *)h0, (u64 *)h0, (size_t)nct * nl, p0));
    hipLaunchKernelGGL(k_add_rows, dim3(N / 256, nl, nct), dim3(256), 0, ctx->stream, (const u64 *)h0, (size_t)nl * N, (const u64 *)mask_ntt, (size_t)nl * N, (u64 *)h0, (size_t)nl * N, ctx->modc);
    hipLaunchKernelGGL(k_share, dim3(N / 256, nl, nct), dim3(256), 0, ctx->stream, (const u64 *)h0, ctx->sh->sk_dev, (const u64 *)ct + (size_t)nl * N, (size_t)2 * nl * N, (u64 *)h0, nl, 0, ctx->modc);
    SFG_HIP(ctx, hipGetLastError());
    return 0;
}

// host big integers for the per-level constants (setup path)
namespace {
struct HBig {
    std::vector<u64> w;
    explicit HBig(u64 v = 0) : w(1, v) {}
    void mul_small(u64 m) { u128 c = 0; for (auto &x : w) { c += (u128)x * m; x = (u64)c; c >>= 64; } if (c) w.push_back((u64)c); }
    u64 divmod_small(u64 d) { u128 r = 0; for (size_t i = w.size(); i-- > 0;) { u128 cur = (r << 64) | w[i]; w[i] = (u64)(cur / d); r = cur % d; } while (w.size() > 1 && !w.back()) w.pop_back(); return (u64)r; }
    void shr1() { for (size_t i = 0; i < w.size(); i++) w[i] = (w[i] >> 1) | (i + 1 < w.size() ? w[i + 1] << 63 : 0); }
};
}
static void recode_constants(const sfg_ctx *ctx, int level, RecodeConst &rc) {
    const int nl = level + 1, nq = ctx->nq;
    memset(&rc, 0, sizeof rc); rc.nl = nl; rc.nq = nq;
    for (int i = 0; i < nl; i++) for (int t = 0; t < i; t++) rc.inv[i][t] = (double)h_invmod(ctx->q[t] % ctx->q[i], ctx->q[i]);
    HBig Q(1); for (int i = 0; i < nl; i++) Q.mul_small(ctx->q[i]);
    HBig H = Q; H.shr1();
    for (int i = 0; i < nl; i++) rc.half[i] = (double)H.divmod_small(ctx->q[i]);
    for (int j = nl; j < nq; j++) {
        u64 qm = 1;
        for (int i = 0; i < nl; i++) { rc.qm[j][i] = (double)(ctx->q[i] % ctx->q[j]); qm = h_mulmod(qm, ctx->q[i] % ctx->q[j], ctx->q[j]); }
        rc.Qmod[j] = (double)qm;
    }
}

static int refresh_finish(sfg_ctx *ctx, const uint64_t *ct, int nct, int level, const uint64_t *h0agg, const uint64_t *h1agg, const uint64_t *crs,
                          uint64_t *out, const ScaleRatio *sr);
extern "C" int sfg_refresh_finish_dev(sfg_ctx *ctx, const uint64_t *ct, int nct, int level, const uint64_t *h0agg, const uint64_t *h1agg, const uint64_t *crs,
                                      uint64_t *out) {
    return refresh_finish(ctx, ct, nct, level, h0agg, h1agg, crs, out, nullptr);
}
extern "C" int sfg_refresh_finish_scaled_dev(sfg_ctx *ctx, const uint64_t *ct, int nct, int level, double ct_scale, double target_scale, const uint64_t *h0agg,
                                             const uint64_t *h1agg, const uint64_t *crs, uint64_t *out) {
    if (level < 0 || level >= ctx->nq) SFG_FAIL(ctx, "refresh: level %d out of range", level);
    if (ct_scale == target_scale) return refresh_finish(ctx, ct, nct, level, h0agg, h1agg, crs, out, nullptr);
    ScaleRatio sr; SFG_TRY(scale_ratio(ctx, level, ct_scale, target_scale, sr));
    return refresh_finish(ctx, ct, nct, level, h0agg, h1agg, crs, out, &sr);
}
static int refresh_finish(sfg_ctx *ctx, const uint64_t *ct, int nct, int level, const uint64_t *h0agg, const uint64_t *h1agg, const uint64_t *crs,
                          uint64_t *out, const ScaleRatio *sr) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    SFG_TRY(refresh_check(ctx, nct, level));
    if (!nct) return 0;
    const int N = SFG_N, nl = level + 1, nq = ctx->nq;
    u64 *x = nullptr;
    SFG_TRY(sfg_scratch(ctx, "refresh.x", (size_t)nct * nl * N * 8, (void **)&x));
    // Decrypt: c0 + h0agg, then to the coefficient domain
    hipLaunchKernelGGL(k_add_rows, dim3(N / 256, nl, nct), dim3(256), 0, ctx->stream, (const u64 *)ct, (size_t)2 * nl * N, (const u64 *)h0agg, (size_t)nl * N, x, (size_t)nl * N, ctx->modc);
    SFG_HIP(ctx, hipGetLastError());
    ModPattern p0; p0.period = nl; for (int j = 0; j < nl; j++) p0.m[j] = (int8_t)j;
    SFG_TRY(launch_ntt_inv(ctx, x, x, (size_t)nct * nl, p0));
    // Recode into all nq moduli
    RecodeConst rc; recode_constants(ctx, level, rc);
    if (sr) {
        RecodeBig rb; memset(&rb, 0, sizeof rb);
        HBig Q(1); for (int i = 0; i < nl; i++) { Q.mul_small(ctx->q[i]); rb.qi[i] = ctx->q[i]; }
        for (size_t i = 0; i < Q.w.size() && i < (size_t)BG; i++) rb.Q[i] = Q.w[i];
        hipLaunchKernelGGL(k_recode_scaled, dim3(N / 256, nct), dim3(256), 0, ctx->stream, (const u64 *)x, (u64 *)out, rc, rb, *sr, ctx->modc);
    } else hipLaunchKernelGGL(k_recode, dim3(N / 256, nct), dim3(256), 0, ctx->stream, (const u64 *)x, (u64 *)out, rc, ctx->modc);
    SFG_HIP(ctx, hipGetLastError());
    ModPattern p1; p1.period = nq; for (int j = 0; j < nq; j++) p1.m[j] = (int8_t)j;
    RowMap rm; rm.rpg = nq; rm.gstride_in = rm.gstride_out = (size_t)2 * nq * N;           // polynomial 0 of every output ciphertext
    SFG_TRY(launch_ntt_fwd_map(ctx, (const u64 *)out, (u64 *)out, (size_t)nct * nq, p1, rm));
    // Recrypt
    hipLaunchKernelGGL(k_recrypt, dim3(N / 256, nq, nct), dim3(256), 0, ctx->stream, (u64 *)out, (const u64 *)h1agg, (const u64 *)crs, nq, ctx->modc);
    SFG_HIP(ctx, hipGetLastError());
    return 0;
}
