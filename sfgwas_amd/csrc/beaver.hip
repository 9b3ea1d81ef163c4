// beaver.hip — local Beaver-triple products (mpc/beavermult.go:94-147) over a prime field whose elements are
// `limbs` little-endian 64-bit words (mpc-core LElem128: limbs = 2, LElem256: limbs = 4; gwas.go:191-199).
//   pid 0     : out = am*bm
//   pid 1     : out = ar*bm + br*am + ar*br
//   pid >= 2  : out = ar*bm + br*am
// Element-wise form (B2) is pure streaming: 4 inputs + 1 output, 16/32 B each, one element per lane, fully
// coalesced, ~HBM-bound.  Field products use Montgomery multiplication on 32-bit words (CIOS); a product is
// mont(mont(a,b), R^2) so inputs and outputs stay in plain (canonical) representation like mpc-core's.
// pid 1 uses ar*(bm+br) + br*am: the same field element with one product fewer.
#include "common.hpp"
#include "kernels.hpp"

struct FieldConst { uint32_t p[8]; uint32_t r2[8]; uint32_t n0inv; int nw; int pm_w, pm_s; uint32_t pm_c; };     // pm_*: p = 2^(32 pm_w + pm_s) - pm_c (pm_w < 0: not of that form)

template <int NW>
__device__ __forceinline__ void f_montmul(const uint32_t (&a)[NW], const uint32_t (&b)[NW], const FieldConst &f, uint32_t (&out)[NW]) {
    uint32_t t[NW + 2];
#pragma unroll
    for (int i = 0; i < NW + 2; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < NW; i++) {
        u64 c = 0;
#pragma unroll
        for (int j = 0; j < NW; j++) { u64 v = (u64)a[j] * b[i] + t[j] + c; t[j] = (uint32_t)v; c = v >> 32; }
        u64 v = (u64)t[NW] + c; t[NW] = (uint32_t)v; t[NW + 1] = (uint32_t)(v >> 32);
        const uint32_t m = t[0] * f.n0inv;
        v = (u64)m * f.p[0] + t[0]; c = v >> 32;
#pragma unroll
        for (int j = 1; j < NW; j++) { v = (u64)m * f.p[j] + t[j] + c; t[j - 1] = (uint32_t)v; c = v >> 32; }
        v = (u64)t[NW] + c; t[NW - 1] = (uint32_t)v; t[NW] = t[NW + 1] + (uint32_t)(v >> 32);
    }
    // conditional subtract: t (NW+1 words) >= p ?
    bool ge = t[NW] != 0;
    if (!ge) {
        ge = true;
#pragma unroll
        for (int j = NW - 1; j >= 0; j--) { if (t[j] != f.p[j]) { ge = t[j] > f.p[j]; break; } }
    }
    u64 br = 0;
#pragma unroll
    for (int j = 0; j < NW; j++) { u64 d = (u64)t[j] - (ge ? f.p[j] : 0) - br; out[j] = (uint32_t)d; br = (d >> 32) & 1; }
}
template <int NW>
__device__ __forceinline__ void f_add(const uint32_t (&a)[NW], const uint32_t (&b)[NW], const FieldConst &f, uint32_t (&out)[NW]) {
    uint32_t s[NW]; u64 c = 0;
#pragma unroll
    for (int j = 0; j < NW; j++) { u64 v = (u64)a[j] + b[j] + c; s[j] = (uint32_t)v; c = v >> 32; }
    bool ge = c != 0;
    if (!ge) {
        ge = true;
#pragma unroll
        for (int j = NW - 1; j >= 0; j--) { if (s[j] != f.p[j]) { ge = s[j] > f.p[j]; break; } }
    }
    u64 br = 0;
#pragma unroll
    for (int j = 0; j < NW; j++) { u64 d = (u64)s[j] - (ge ? f.p[j] : 0) - br; out[j] = (uint32_t)d; br = (d >> 32) & 1; }
}
template <int NW>
__device__ __forceinline__ void f_mul(const uint32_t (&a)[NW], const uint32_t (&b)[NW], const FieldConst &f, uint32_t (&out)[NW]) {
    uint32_t t[NW], r2[NW];
#pragma unroll
    for (int j = 0; j < NW; j++) r2[j] = f.r2[j];
    f_montmul<NW>(a, b, f, t);
    f_montmul<NW>(t, r2, f, out);
}
template <int NW>
__device__ __forceinline__ void f_load(const uint64_t *p, size_t e, uint32_t (&x)[NW]) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p + e * (NW / 2));
#pragma unroll
    for (int k = 0; k < NW / 4; k++) { uint4 v = q[k]; x[4 * k] = v.x; x[4 * k + 1] = v.y; x[4 * k + 2] = v.z; x[4 * k + 3] = v.w; }
}
template <int NW>
__device__ __forceinline__ void f_store(uint64_t *p, size_t e, const uint32_t (&x)[NW]) {
    uint4 *q = reinterpret_cast<uint4 *>(p + e * (NW / 2));
#pragma unroll
    for (int k = 0; k < NW / 4; k++) q[k] = make_uint4(x[4 * k], x[4 * k + 1], x[4 * k + 2], x[4 * k + 3]);
}

// ---- pseudo-Mersenne moduli p = 2^B - c, c < 2^32 (both candidates for mpc-core's fields are: 2^127 - 1, 2^255 - 19): the whole share
// expression a1*b1 + a2*b2 is formed as ONE double-width integer (2 NW^2 multiply-adds) and reduced by folding the part above 2^B back as c * H - three
// folds of at most NW + 3 single-word products and one conditional subtraction - instead of two Montgomery products (4 NW^2) per field product.
// Same canonical residues; the generic Montgomery path remains for any other odd modulus.  W = B / 32, s = B % 32.
template <int NIN, int W, int NOUT>
__device__ __forceinline__ void pm_fold(const uint32_t (&in)[NIN], int s, uint32_t c, uint32_t (&out)[NOUT]) {
    constexpr int NH = NIN - W;                                        // words of H = in >> B
    static_assert(NOUT >= W + 1, "fold output too narrow");
    const uint32_t lowmask = s ? ((1u << s) - 1u) : 0u;
    u64 carry = 0;
#pragma unroll
    for (int k = 0; k < NOUT; k++) {
        const uint32_t lo = k < W ? in[k] : (k == W ? in[W] & lowmask : 0u);
        uint32_t h = 0;
        if (k < NH) h = __funnelshift_r(in[W + k], W + k + 1 < NIN ? in[W + k + 1] : 0u, s);
        const u64 v = (u64)h * c + lo + carry;
        out[k] = (uint32_t)v; carry = v >> 32;
    }
}
template <int NW, int W, bool TWO>
__device__ __forceinline__ void f_mulsum_pm(const uint32_t (&a1)[NW], const uint32_t (&b1)[NW], const uint32_t (&a2)[NW], const uint32_t (&b2)[NW],
                                            const FieldConst &f, uint32_t (&out)[NW]) {
    uint32_t T[2 * NW + 1];
#pragma unroll
    for (int i = 0; i < 2 * NW + 1; i++) T[i] = 0;
#pragma unroll
    for (int pass = 0; pass < (TWO ? 2 : 1); pass++) {
        const uint32_t (&a)[NW] = pass ? a2 : a1; const uint32_t (&b)[NW] = pass ? b2 : b1;
#pragma unroll
        for (int i = 0; i < NW; i++) {
            u64 c = 0;
#pragma unroll
            for (int j = 0; j < NW; j++) { const u64 v = (u64)a[j] * b[i] + T[i + j] + c; T[i + j] = (uint32_t)v; c = v >> 32; }
#pragma unroll
            for (int j = i + NW; j < 2 * NW + 1; j++) { const u64 v = (u64)T[j] + c; T[j] = (uint32_t)v; c = v >> 32; }
        }
    }
    const int sh = f.pm_s; const uint32_t c = f.pm_c;
    uint32_t S1[NW + 4], S2[W + 2], S3[W + 1];
    pm_fold<2 * NW + 1, W, NW + 4>(T, sh, c, S1);                       // < 2^B + c 2^(32 (2 NW + 1) - B)
    pm_fold<NW + 4, W, W + 2>(S1, sh, c, S2);                            // < 2^B + 2^(32 (NW + 4 - W) + 32)
    pm_fold<W + 2, W, W + 1>(S2, sh, c, S3);                             // < 2^B <= p + c
    uint32_t t[NW];
#pragma unroll
    for (int j = 0; j < NW; j++) t[j] = j < W + 1 ? S3[j] : 0u;
    bool ge = true;
#pragma unroll
    for (int j = NW - 1; j >= 0; j--) { if (t[j] != f.p[j]) { ge = t[j] > f.p[j]; break; } }
    u64 br = 0;
#pragma unroll
    for (int j = 0; j < NW; j++) { const u64 d = (u64)t[j] - (ge ? f.p[j] : 0) - br; out[j] = (uint32_t)d; br = (d >> 32) & 1; }
}

// BeaverMultElemMat (beavermult.go:112-133).  W >= 0 selects the pseudo-Mersenne reduction with B / 32 = W.
template <int NW, int W>
__global__ void __launch_bounds__(256) k_beaver_elem_pm(int pid, FieldConst f, const uint64_t *ar, const uint64_t *am, const uint64_t *br, const uint64_t *bm,
                                                       uint64_t *out, size_t n) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        uint32_t xam[NW], xbm[NW], o[NW];
        f_load<NW>(am, e, xam); f_load<NW>(bm, e, xbm);
        if (pid == 0) f_mulsum_pm<NW, W, false>(xam, xbm, xam, xbm, f, o);
        else {
            uint32_t xar[NW], xbr[NW], t[NW];
            f_load<NW>(ar, e, xar); f_load<NW>(br, e, xbr);
            if (pid == 1) { f_add<NW>(xbm, xbr, f, t); f_mulsum_pm<NW, W, true>(xar, t, xbr, xam, f, o); }      // ar*(bm + br) + br*am
            else f_mulsum_pm<NW, W, true>(xar, xbm, xbr, xam, f, o);                                          // ar*bm + br*am
        }
        f_store<NW>(out, e, o);
    }
}
template <int NW>
__global__ void __launch_bounds__(256) k_beaver_elem(int pid, FieldConst f, const uint64_t *ar, const uint64_t *am, const uint64_t *br, const uint64_t *bm,
                                                    uint64_t *out, size_t n) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        uint32_t xam[NW], xbm[NW], o[NW];
        f_load<NW>(am, e, xam); f_load<NW>(bm, e, xbm);
        if (pid == 0) { f_mul<NW>(xam, xbm, f, o); }
        else {
            uint32_t xar[NW], xbr[NW], t[NW], u[NW];
            f_load<NW>(ar, e, xar); f_load<NW>(br, e, xbr);
            if (pid == 1) { f_add<NW>(xbm, xbr, f, t); f_mul<NW>(xar, t, f, u); }      // ar*(bm + br)
            else f_mul<NW>(xar, xbm, f, u);                                           // ar*bm
            f_mul<NW>(xbr, xam, f, t);                                                // br*am
            f_add<NW>(u, t, f, o);
        }
        f_store<NW>(out, e, o);
    }
}
// BeaverMultMat (beavermult.go:135-147): out[m x n] from [m x k] and [k x n] operands; one thread per output element
template <int NW>
__global__ void __launch_bounds__(64) k_beaver_matmul(int pid, FieldConst f, const uint64_t *ar, const uint64_t *am, const uint64_t *br, const uint64_t *bm,
                                                     uint64_t *out, int m, int k, int n) {
    const int idx = blockIdx.x * 64 + threadIdx.x;
    if (idx >= m * n) return;
    const int i = idx / n, j = idx % n;
    uint32_t acc[NW];
#pragma unroll
    for (int w = 0; w < NW; w++) acc[w] = 0;
    for (int x = 0; x < k; x++) {
        uint32_t a1[NW], b1[NW], t[NW], u[NW];
        if (pid == 0) { f_load<NW>(am, (size_t)i * k + x, a1); f_load<NW>(bm, (size_t)x * n + j, b1); f_mul<NW>(a1, b1, f, t); }
        else {
            uint32_t a2[NW], b2[NW];
            f_load<NW>(ar, (size_t)i * k + x, a1); f_load<NW>(bm, (size_t)x * n + j, b1);
            f_load<NW>(am, (size_t)i * k + x, a2); f_load<NW>(br, (size_t)x * n + j, b2);
            if (pid == 1) { f_add<NW>(b1, b2, f, u); f_mul<NW>(a1, u, f, t); } else f_mul<NW>(a1, b1, f, t);   // ar*bm (+ ar*br)
            f_mul<NW>(a2, b2, f, u); f_add<NW>(t, u, f, t);                                                    // + am*br
        }
        f_add<NW>(acc, t, f, acc);
    }
    f_store<NW>(out, (size_t)idx, acc);
}

// host big-integer helpers (setup only): R^2 mod p and -p^-1 mod 2^32
static int field_setup(sfg_ctx *ctx, int limbs, const uint64_t *mod, FieldConst &f) {
    if (limbs != 2 && limbs != 4) SFG_FAIL(ctx, "beaver: limbs must be 2 (128-bit) or 4 (256-bit)");
    const int nw = 2 * limbs; f.nw = nw;
    for (int j = 0; j < 8; j++) { f.p[j] = 0; f.r2[j] = 0; }
    for (int j = 0; j < limbs; j++) { f.p[2 * j] = (uint32_t)mod[j]; f.p[2 * j + 1] = (uint32_t)(mod[j] >> 32); }
    if (!(f.p[0] & 1)) SFG_FAIL(ctx, "beaver: modulus must be odd");
    uint32_t inv = 1; for (int i = 0; i < 5; i++) inv *= 2 - f.p[0] * inv;          // p^-1 mod 2^32 (Newton)
    f.n0inv = (uint32_t)(0u - inv);
    // R^2 mod p with R = 2^(32 nw): start from 1 and double 2*32*nw times modulo p
    uint32_t x[9] = {1, 0, 0, 0, 0, 0, 0, 0, 0};
    auto ge = [&](const uint32_t *a) { if (a[nw]) return true; for (int j = nw - 1; j >= 0; j--) if (a[j] != f.p[j]) return a[j] > f.p[j]; return true; };
    for (int it = 0; it < 64 * nw; it++) {
        uint32_t c = 0;
        for (int j = 0; j <= nw; j++) { uint32_t nc = x[j] >> 31; x[j] = (x[j] << 1) | c; c = nc; }
        if (ge(x)) { u64 br = 0; for (int j = 0; j <= nw; j++) { u64 d = (u64)x[j] - (j < nw ? f.p[j] : 0) - br; x[j] = (uint32_t)d; br = (d >> 32) & 1; } }
    }
    for (int j = 0; j < nw; j++) f.r2[j] = x[j];
    // p = 2^B - c with c < 2^32 and the top word in use?  (c = 2^B - p: every word above the lowest must be all ones up to bit B)
    f.pm_w = -1; f.pm_s = 0; f.pm_c = 0;
    if (f.p[nw - 1]) {
        int B = 32 * nw; while (!((f.p[(B - 1) / 32] >> ((B - 1) % 32)) & 1)) B--;
        bool ones = true;
        for (int b = 32; b < B && ones; b++) ones = (f.p[b / 32] >> (b % 32)) & 1;
        if (ones && B > 32 * (nw - 1)) { f.pm_w = B / 32; f.pm_s = B % 32; f.pm_c = (uint32_t)(0u - f.p[0]); }
    }
    return 0;
}

extern "C" int sfg_beaver_elem_dev(sfg_ctx *ctx, int pid, int limbs, const uint64_t *mod, const uint64_t *ar, const uint64_t *am,
                                   const uint64_t *br, const uint64_t *bm, uint64_t *out, size_t n) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (!n) return 0;
    FieldConst f; SFG_TRY(field_setup(ctx, limbs, mod, f));
    size_t blocks = (n + 255) / 256; if (blocks > 8192) blocks = 8192;
    const dim3 g((unsigned)blocks), b(256);
    if (limbs == 2 && f.pm_w == 3) hipLaunchKernelGGL((k_beaver_elem_pm<4, 3>), g, b, 0, ctx->stream, pid, f, ar, am, br, bm, out, n);
    else if (limbs == 2 && f.pm_w == 4) hipLaunchKernelGGL((k_beaver_elem_pm<4, 4>), g, b, 0, ctx->stream, pid, f, ar, am, br, bm, out, n);
    else if (limbs == 4 && f.pm_w == 7) hipLaunchKernelGGL((k_beaver_elem_pm<8, 7>), g, b, 0, ctx->stream, pid, f, ar, am, br, bm, out, n);
    else if (limbs == 4 && f.pm_w == 8) hipLaunchKernelGGL((k_beaver_elem_pm<8, 8>), g, b, 0, ctx->stream, pid, f, ar, am, br, bm, out, n);
    else if (limbs == 2) hipLaunchKernelGGL(k_beaver_elem<4>, g, b, 0, ctx->stream, pid, f, ar, am, br, bm, out, n);
    else hipLaunchKernelGGL(k_beaver_elem<8>, g, b, 0, ctx->stream, pid, f, ar, am, br, bm, out, n);
    SFG_HIP(ctx, hipGetLastError());
    return 0;
}

static int beaver_host_common(sfg_ctx *ctx, int pid, int limbs, const uint64_t *mod, const uint64_t *ar, const uint64_t *am, const uint64_t *br,
                              const uint64_t *bm, uint64_t *out, size_t na, size_t nb, size_t nout, int m, int k, int n) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    const size_t ea = na * limbs * 8, eb = nb * limbs * 8, eo = nout * limbs * 8;
    uint64_t *d = nullptr;
    SFG_HIP(ctx, hipMalloc(&d, 2 * ea + 2 * eb + eo + 64));
    uint64_t *dar = d, *dam = d + na * limbs, *dbr = dam + na * limbs, *dbm = dbr + nb * limbs, *dout = dbm + nb * limbs;
    int rc = 0;
    auto up = [&](uint64_t *dst, const uint64_t *src, size_t bytes) { if (src && hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = 1; };
    if (pid != 0) { up(dar, ar, ea); up(dbr, br, eb); }
    else { (void)hipMemsetAsync(dar, 0, ea, ctx->stream); (void)hipMemsetAsync(dbr, 0, eb, ctx->stream); }
    up(dam, am, ea); up(dbm, bm, eb);
    if (rc) { (void)hipFree(d); SFG_FAIL(ctx, "beaver: upload failed"); }
    if (m < 0) rc = sfg_beaver_elem_dev(ctx, pid, limbs, mod, dar, dam, dbr, dbm, dout, nout);
    else {
        FieldConst f; rc = field_setup(ctx, limbs, mod, f);
        if (!rc) {
            const int blocks = (m * n + 63) / 64;
            if (limbs == 2) hipLaunchKernelGGL(k_beaver_matmul<4>, dim3(blocks), dim3(64), 0, ctx->stream, pid, f, dar, dam, dbr, dbm, dout, m, k, n);
            else hipLaunchKernelGGL(k_beaver_matmul<8>, dim3(blocks), dim3(64), 0, ctx->stream, pid, f, dar, dam, dbr, dbm, dout, m, k, n);
            if (hipGetLastError() != hipSuccess) { rc = 1; ctx->err = "beaver_matmul launch failed"; }
        }
    }
    if (!rc && hipMemcpyAsync(out, dout, eo, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) { rc = 1; ctx->err = "beaver: download failed"; }
    (void)hipStreamSynchronize(ctx->stream); (void)hipFree(d);
    return rc;
}
extern "C" int sfg_beaver_elem(sfg_ctx *ctx, int pid, int limbs, const uint64_t *mod, const uint64_t *ar, const uint64_t *am, const uint64_t *br,
                               const uint64_t *bm, uint64_t *out, size_t n) {
    return beaver_host_common(ctx, pid, limbs, mod, ar, am, br, bm, out, n, n, n, -1, 0, 0);
}
extern "C" int sfg_beaver_matmul(sfg_ctx *ctx, int pid, int limbs, const uint64_t *mod, const uint64_t *ar, const uint64_t *am, const uint64_t *br,
                                 const uint64_t *bm, uint64_t *out, int m, int k, int n) {
    if (m < 1 || k < 1 || n < 1) SFG_FAIL(ctx, "beaver_matmul: bad dimensions");
    return beaver_host_common(ctx, pid, limbs, mod, ar, am, br, bm, out, (size_t)m * k, (size_t)k * n, (size_t)m * n, m, k, n);
}

// ---------------------------------------------------------------- f-4: the share algebra of MPC.SSToCMat that does not need the fork (mpc/ss.go:84-110)
//   mask[i][j] = FromBigInt(tmp); if tmp >= bound / 2: mask -= FromBigInt(bound)          (:90-99; tmp = ring.RandInt(bound) stays with the caller)
//   rmMask = rm - mask                                                                      (:101-102, before RevealSymMat)
//   share  = hub ? revealed + mask : mask                                                   (:104-110, after RevealSymMat)
// Field elements as in the Beaver products: `limbs` little-endian 64-bit words, any odd modulus.  One element per lane, streaming (16 / 32 B per operand).
// mode 0: out = a - recentre(b)   (a = rm, b = raw mask tmp < bound)      mask_out (nullable) = recentre(b)
// mode 1: out = a + b             (a = revealed, b = mask)
template <int NW>
__global__ void __launch_bounds__(256) k_ss_share(int mode, FieldConst f, FieldConst bnd /* p = bound, r2 = bound / 2 */, const uint64_t *a, const uint64_t *b, uint64_t *out, uint64_t *mask_out, size_t n) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) {
        uint32_t x[NW], y[NW], r[NW];
        const uint32_t *pa = reinterpret_cast<const uint32_t *>(a) + e * NW, *pb = reinterpret_cast<const uint32_t *>(b) + e * NW;
#pragma unroll
        for (int j = 0; j < NW; j++) { x[j] = pa[j]; y[j] = pb[j]; }
        if (mode == 0) {
            bool ge = true;                                            // tmp.Cmp(boundHalf) >= 0
#pragma unroll
            for (int j = NW - 1; j >= 0; j--) { if (y[j] != bnd.r2[j]) { ge = y[j] > bnd.r2[j]; break; } }
            if (ge) {                                                  // mask = tmp - bound mod p  =  p - (bound - tmp)   (bound - tmp in (0, bound / 2] < p)
                uint32_t d[NW]; u64 br = 0;
#pragma unroll
                for (int j = 0; j < NW; j++) { u64 v = (u64)bnd.p[j] - y[j] - br; d[j] = (uint32_t)v; br = (v >> 32) & 1; }
                br = 0;
#pragma unroll
                for (int j = 0; j < NW; j++) { u64 v = (u64)f.p[j] - d[j] - br; y[j] = (uint32_t)v; br = (v >> 32) & 1; }
            }
            if (mask_out) { uint32_t *pm = reinterpret_cast<uint32_t *>(mask_out) + e * NW;
#pragma unroll
                for (int j = 0; j < NW; j++) pm[j] = y[j]; }
            // x - y mod p
            u64 br = 0;
#pragma unroll
            for (int j = 0; j < NW; j++) { u64 v = (u64)x[j] - y[j] - br; r[j] = (uint32_t)v; br = (v >> 32) & 1; }
            if (br) { u64 c = 0;
#pragma unroll
                for (int j = 0; j < NW; j++) { u64 v = (u64)r[j] + f.p[j] + c; r[j] = (uint32_t)v; c = v >> 32; } }
        } else f_add<NW>(x, y, f, r);
        uint32_t *po = reinterpret_cast<uint32_t *>(out) + e * NW;
#pragma unroll
        for (int j = 0; j < NW; j++) po[j] = r[j];
    }
}
static int ss_share(sfg_ctx *ctx, int mode, int limbs, const uint64_t *mod, const uint64_t *bound, const uint64_t *a, const uint64_t *b, uint64_t *out, uint64_t *mask_out, size_t n) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (!n) return 0;
    FieldConst f; SFG_TRY(field_setup(ctx, limbs, mod, f));
    FieldConst bd; memset(&bd, 0, sizeof bd);
    if (mode == 0) {
        if (!bound) SFG_FAIL(ctx, "ss_share: no bound");
        for (int j = 0; j < limbs; j++) { bd.p[2 * j] = (uint32_t)bound[j]; bd.p[2 * j + 1] = (uint32_t)(bound[j] >> 32); }
        for (int j = 0; j < limbs; j++) { const uint64_t h = (bound[j] >> 1) | (j + 1 < limbs ? bound[j + 1] << 63 : 0); bd.r2[2 * j] = (uint32_t)h; bd.r2[2 * j + 1] = (uint32_t)(h >> 32); }
        for (int j = limbs - 1; j >= 0; j--) { if (bound[j] != mod[j]) { if (bound[j] > mod[j]) SFG_FAIL(ctx, "ss_share: bound exceeds the field modulus"); break; } }
    }
    size_t blocks = (n + 255) / 256; if (blocks > 8192) blocks = 8192;
    if (limbs == 2) hipLaunchKernelGGL(k_ss_share<4>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, mode, f, bd, a, b, out, mask_out, n);
    else hipLaunchKernelGGL(k_ss_share<8>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, mode, f, bd, a, b, out, mask_out, n);
    SFG_HIP(ctx, hipGetLastError());
    return 0;
}
extern "C" int sfg_ss_mask_dev(sfg_ctx *ctx, int limbs, const uint64_t *modulus_host, const uint64_t *bound_host, const uint64_t *rm_dev, const uint64_t *rand_dev,
                               uint64_t *rm_masked_dev, uint64_t *mask_dev, size_t n) {
    return ss_share(ctx, 0, limbs, modulus_host, bound_host, rm_dev, rand_dev, rm_masked_dev, mask_dev, n);
}
extern "C" int sfg_ss_hub_share_dev(sfg_ctx *ctx, int limbs, const uint64_t *modulus_host, const uint64_t *revealed_dev, const uint64_t *mask_dev, uint64_t *share_dev, size_t n) {
    return ss_share(ctx, 1, limbs, modulus_host, nullptr, revealed_dev, mask_dev, share_dev, nullptr, n);
}
