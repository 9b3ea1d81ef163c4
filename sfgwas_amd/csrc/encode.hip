// encode.hip — generalized-diagonal extraction and CKKS encoding of int8 genotype blocks
// (GetDiag / convertToComplex128WithRot / EncodeDiagWithEncoder -> lattigo EncoderBig.EncodeNTT,
//  gwas/matmult.go:636-731; called per diagonal at :1024 and :1426).
//
// The reference encodes with 256-bit big floats, i.e. it produces the exactly rounded integers
//     p_c = round(Delta * Re w_c),  p_{c+n} = round(Delta * Im w_c),   w_c = (1/n) sum_t v_t zeta^(-5^t c)
// (n = N/2 slots, zeta = exp(2 pi i / 2N)).  Here the same real numbers are computed in double-double
// arithmetic (~2^-104 relative) and rounded half away from zero, which gives the same integers.
//   * u[(5^t - 1)/4 mod n] = v_t turns the sum into an ordinary length-n DFT times zeta^(-c);
//   * v is real, so the DFT is done as a length-n/2 complex FFT of z_m = u_2m + i u_2m+1 plus one
//     recombination pass, and only p_0..p_{n-1} are produced: p_n = 0 and p_{N-c} = -p_c follow from
//     invariance under X -> X^-1 (the NTT kernel expands them, ntt.hip IN_MODE 1).
// One 512-thread workgroup encodes one diagonal; the 4096-point FFT runs as 4 radix-8 register passes whose exchanges
// move the high parts of all points through a 64 KiB LDS image (re, im), then the low parts.
#include "common.hpp"
#include "kernels.hpp"
#include "i8_move.hpp"            // PtRide: the riding transposition's items are dealt out over the NTT launches made here
#include <cmath>
#include <type_traits>

// ---------------------------------------------------------------- double-double (host + device)
struct dd { double hi, lo; };
__host__ __device__ static inline dd dd_make(double h, double l) { dd r; r.hi = h; r.lo = l; return r; }
__host__ __device__ static inline dd dd_quick(double a, double b) { double s = a + b; return dd_make(s, b - (s - a)); }
__host__ __device__ static inline dd dd_two_sum(double a, double b) { double s = a + b, bb = s - a; return dd_make(s, (a - (s - bb)) + (b - bb)); }
__host__ __device__ static inline dd dd_add(dd a, dd b) { dd s = dd_two_sum(a.hi, b.hi); s.lo += a.lo + b.lo; return dd_quick(s.hi, s.lo); }
__host__ __device__ static inline dd dd_neg(dd a) { return dd_make(-a.hi, -a.lo); }
// sum without the final renormalisation (8 flops instead of 11): the high parts are added exactly, the low part may grow to a
// few ulps of the high part.  Every consumer below (two_sum on the high parts, dd_mul / dd_dot2 cross terms) is exact or
// first-order correct for such pairs, so a chain of k lazy sums costs log2(k) of the ~106 bits; results are renormalised by the
// next product or by the recombination.
__host__ __device__ static inline dd dd_add_lazy(dd a, dd b) { dd s = dd_two_sum(a.hi, b.hi); s.lo += a.lo + b.lo; return s; }
__host__ __device__ static inline dd dd_sub_lazy(dd a, dd b) { dd s = dd_two_sum(a.hi, -b.hi); s.lo += a.lo - b.lo; return s; }
__host__ __device__ static inline dd dd_sub(dd a, dd b) { return dd_add(a, dd_neg(b)); }
__host__ __device__ static inline dd dd_mul(dd a, dd b) {
    double p = a.hi * b.hi, e = fma(a.hi, b.hi, -p);
    e = fma(a.hi, b.lo, e); e = fma(a.lo, b.hi, e);
    return dd_quick(p, e);
}
__host__ __device__ static inline dd dd_mul_d(dd a, double b) {
    double p = a.hi * b, e = fma(a.hi, b, -p);
    e = fma(a.lo, b, e);
    return dd_quick(p, e);
}
// accurate host-only add (table construction)
static inline dd dd_add_acc(dd a, dd b) {
    dd s = dd_two_sum(a.hi, b.hi), t = dd_two_sum(a.lo, b.lo);
    s.lo += t.hi; s = dd_quick(s.hi, s.lo); s.lo += t.lo; return dd_quick(s.hi, s.lo);
}
struct cdd { dd re, im; };
__host__ __device__ static inline cdd cdd_mul(cdd a, cdd b) {
    cdd r; r.re = dd_sub(dd_mul(a.re, b.re), dd_mul(a.im, b.im)); r.im = dd_add(dd_mul(a.re, b.im), dd_mul(a.im, b.re)); return r;
}

constexpr int ENC_H = SFG_SLOTS / 2;         // 4096-point complex FFT
constexpr int ENC_TW = 16384;                // zeta^-k is built for k = 0..16384, zeta = exp(2 pi i / 32768)
// device twiddle table (double4 entries): the three twiddled radix-8 passes, then the recombination lists
constexpr int ENC_TB_P512 = 0, ENC_TB_P64 = ENC_TB_P512 + 7 * 512, ENC_TB_P8 = ENC_TB_P64 + 7 * 64;
constexpr int ENC_TB_RLEN = ENC_H / 2 + 1;   // c = 0..h/2
constexpr int ENC_TB_RW = ENC_TB_P8 + 7 * 8, ENC_TB_RZ = ENC_TB_RW + ENC_TB_RLEN, ENC_TB_RZ2 = ENC_TB_RZ + ENC_TB_RLEN;
constexpr int ENC_TB_SIZE = ENC_TB_RZ2 + ENC_TB_RLEN;

struct EncTables {                // immutable, shared by a context and its forks
    double4 *tb = nullptr;        // twiddles {re.hi, re.lo, im.hi, im.lo} of zeta^-k = exp(-2 pi i k / 32768), laid out in the order the kernel's lanes read them (ENC_TB_*)
    uint16_t *tinv = nullptr;     // [n] slot index t with (5^t - 1)/4 mod n == m
    double2 *costab = nullptr;    // [8193] cos(2 pi k / 32768) as {hi, lo}: the exact re-derivation of a coefficient next to a rounding tie (k_fft_encode)
    int sexp = -1;                // log2(Delta / n) when that is a power of two (every preset), else -1: no re-derivation
};
// per-context scratch (sfg_scratch pool): "enc.skew" diag-major copy of one block [n][n] int8;
// "enc.pc" half-coefficient rows [batch][n]: the rounded integers, held as doubles (|p| < 2^53)

// cos/sin(theta) for small theta by Taylor series in double-double
static void dd_sincos_small(dd theta, dd &s, dd &c) {
    dd t2 = dd_mul(theta, theta);
    dd term = theta; s = theta;
    for (int k = 1; k < 14; k++) {            // sin: term *= -t2 / ((2k)(2k+1))
        term = dd_mul(term, t2); term = dd_mul_d(term, -1.0);
        double den = (double)(2 * k) * (double)(2 * k + 1);
        // divide by an exactly representable small integer: one Newton-free step via hi/lo correction
        dd q; q.hi = term.hi / den; double rem = fma(-q.hi, den, term.hi); q.lo = (rem + term.lo) / den; term = dd_quick(q.hi, q.lo);
        s = dd_add_acc(s, term);
    }
    term = dd_make(1.0, 0.0); c = term;
    for (int k = 1; k < 14; k++) {            // cos: term *= -t2 / ((2k-1)(2k))
        term = dd_mul(term, t2); term = dd_mul_d(term, -1.0);
        double den = (double)(2 * k - 1) * (double)(2 * k);
        dd q; q.hi = term.hi / den; double rem = fma(-q.hi, den, term.hi); q.lo = (rem + term.lo) / den; term = dd_quick(q.hi, q.lo);
        c = dd_add_acc(c, term);
    }
}

int sfg_encoder_init(sfg_ctx *ctx) {
    EncTables *et = new EncTables();
    ctx->sh->enc_tables = et;
    const int n = SFG_SLOTS; const u64 M = 2ULL * SFG_N;
    // zeta^-1 = exp(-i theta), theta = 2 pi / 32768 = pi * 2^-14 (exact scaling of the dd constant pi)
    dd pi = dd_make(3.141592653589793116e+00, 1.224646799147353207e-16);
    dd theta = dd_make(pi.hi / 16384.0, pi.lo / 16384.0);
    dd s1, c1; dd_sincos_small(theta, s1, c1);
    std::vector<cdd> z(ENC_TW + 1);
    z[0].re = dd_make(1, 0); z[0].im = dd_make(0, 0);
    z[1].re = c1; z[1].im = dd_neg(s1);
    for (int k = 2; k <= ENC_TW; k++) z[k] = (k & 1) ? cdd_mul(z[k - 1], z[1]) : cdd_mul(z[k / 2], z[k / 2]);
    // exact values at the octants
    z[ENC_TW].re = dd_make(-1, 0); z[ENC_TW].im = dd_make(0, 0);                    // k = 16384: exp(-i pi)
    z[ENC_TW / 2].re = dd_make(0, 0); z[ENC_TW / 2].im = dd_make(-1, 0);            // k = 8192: exp(-i pi/2)
    // per-pass tables: entry [r - 1][t] = W_{8S}^(bitrev3(r) t) = zeta^-(bitrev3(r) t 4096 / S), so lane t of a wave reads consecutive 32-byte entries;
    // recombination tables for c = 0..h/2: omega^-c = zeta^-4c, zeta^-c, zeta^-(h - c)
    auto zat = [&](int idx) {                                   // idx in [0, 32768): the second half is the negated first half
        const bool neg = idx > ENC_TW; const cdd w = z[neg ? idx - ENC_TW : idx]; const double sg = neg ? -1.0 : 1.0;
        return make_double4(sg * w.re.hi, sg * w.re.lo, sg * w.im.hi, sg * w.im.lo);
    };
    std::vector<double4> tb(ENC_TB_SIZE);
    {
        const int E[8] = {0, 4, 2, 6, 1, 5, 3, 7};
        const int S3[3] = {512, 64, 8}, off[3] = {ENC_TB_P512, ENC_TB_P64, ENC_TB_P8};
        for (int k = 0; k < 3; k++) for (int r = 1; r < 8; r++) for (int t = 0; t < S3[k]; t++) tb[off[k] + (r - 1) * S3[k] + t] = zat(E[r] * t * (4096 / S3[k]));
        // the final twist carries the scaling Delta / n as well (exact when that is a power of two, as with every preset; one double-double product otherwise)
        const double son = ctx->sh->scale / (double)n;
        auto scaled = [&](int idx) {
            const double4 w = zat(idx); const dd re = dd_mul_d(dd_make(w.x, w.y), son), im = dd_mul_d(dd_make(w.z, w.w), son);
            return make_double4(re.hi, re.lo, im.hi, im.lo);
        };
        for (int c = 0; c < ENC_TB_RLEN; c++) { { const double4 w = zat(4 * c); tb[ENC_TB_RW + c] = make_double4(0.5 * w.x, 0.5 * w.y, 0.5 * w.z, 0.5 * w.w); } tb[ENC_TB_RZ + c] = scaled(c); tb[ENC_TB_RZ2 + c] = scaled(ENC_H - c); }
    }
    std::vector<uint16_t> tinv(n);
    u64 g = 1;
    for (int t = 0; t < n; t++) { tinv[((g - 1) / 4) % n] = (uint16_t)t; g = (g * 5) % M; }
    {   // quarter-wave cosine table for the near-tie re-derivation: cos(2 pi k / 32768) = Re zeta^-k
        std::vector<double2> ct(ENC_TW / 2 + 1);
        for (int k = 0; k <= ENC_TW / 2; k++) ct[k] = make_double2(z[k].re.hi, z[k].re.lo);
        SFG_HIP(ctx, hipMalloc(&et->costab, ct.size() * sizeof(double2)));
        SFG_HIP(ctx, hipMemcpy(et->costab, ct.data(), ct.size() * sizeof(double2), hipMemcpyHostToDevice));
        int e = 0; const double son = std::frexp(ctx->sh->scale / (double)n, &e);
        et->sexp = (son == 0.5 && e - 1 >= 0 && e - 1 <= 40) ? e - 1 : -1;
    }
    SFG_HIP(ctx, hipMalloc(&et->tb, tb.size() * sizeof(double4)));
    SFG_HIP(ctx, hipMalloc(&et->tinv, n * sizeof(uint16_t)));
    SFG_HIP(ctx, hipMemcpy(et->tb, tb.data(), tb.size() * sizeof(double4), hipMemcpyHostToDevice));
    SFG_HIP(ctx, hipMemcpy(et->tinv, tinv.data(), n * sizeof(uint16_t), hipMemcpyHostToDevice));
    return 0;
}
void sfg_encoder_destroy(SfgShared *sh) {
    EncTables *et = (EncTables *)sh->enc_tables;
    if (!et) return;
    (void)hipFree(et->tb); (void)hipFree(et->tinv); (void)hipFree(et->costab);
    delete et; sh->enc_tables = nullptr;
}

// ---------------------------------------------------------------- diagonal-major copy of one block
// D[shift][j] = X[(shift + j) mod n][j] inside the r x c block, 0 outside (GetDiag, matmult.go:636-664 with
// index = -shift).  A diagonal that "does not exist" (GetDiagBool false) is all zero here, which encodes to
// the zero plaintext — the same contribution as the reference's skipped nil plaintext (matmult.go:392).
// Tiled: a workgroup produces 128 shifts x 128 columns.  The X elements it needs, X[(S0+J0+u) mod n][J0+jj] with
// u = s + jj < 255, are 255 row segments of 128 contiguous bytes (for the transposed operand: 128 rows of 255
// contiguous bytes), staged through LDS so that both the HBM reads and the D writes are contiguous runs.
// The tile is read back as DWORDS and the diagonal bytes are gathered with v_perm_b32: a thread produces 4 shifts x 4 columns from 7 dword reads
// (10 byte-permutes; transposed operand: 7 reads, 3 funnel shifts, 8 permutes) instead of 16 byte reads, and missing -> 0 is applied to 4 packed bytes at once.
constexpr int SK_T = 128, SK_U = 2 * SK_T - 1, SK_PITCH = 144, SK_PITCH_T = 272;      // pitches: multiples of 16 bytes, odd multiples of 4 dwords + ... (dword stride 4 * pitch / 4 + 1 = 17 mod 64: conflict-free gathers)
constexpr int SK_LDS = SK_U * SK_PITCH > SK_T * SK_PITCH_T ? SK_U * SK_PITCH : SK_T * SK_PITCH_T;
// genotype bytes -> contribution bytes: missing (negative) -> 0 (matmult.go:1292-1295), optional squaring in int8 arithmetic (:1301-1303), 4 at a time
__device__ __forceinline__ unsigned sk_clean(unsigned w, int square) {
    const unsigned t = w & 0x80808080u;
    w &= ~((t << 1) - (t >> 7));                          // 0xFF in every byte whose sign bit is set (no borrow crosses a byte: each term is 0x100 - 0x01)
    if (square) {
        unsigned o = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) { const unsigned v = (w >> (8 * k)) & 0xFFu; o |= ((v * v) & 0xFFu) << (8 * k); }
        w = o;
    }
    return w;
}
template <bool TR>
__global__ void __launch_bounds__(256) k_skew(const int8_t *blk, size_t ld, int r, int c, int square, int8_t *D) {
    __shared__ __attribute__((aligned(16))) int8_t tile[SK_LDS];          // tile[u][jj] (pitch 132); transposed operand: tile[jj][u] (pitch 260)
    const int n = SFG_SLOTS, tid = threadIdx.x;
    const int J0 = blockIdx.x * SK_T, S0 = blockIdx.y * SK_T;
    // 4 consecutive bytes along the contiguous axis: one dword load when the run is inside the block and 4-byte aligned
    auto load4 = [&](const int8_t *src, int nvalid) -> unsigned {
        if (nvalid >= 4 && ((uintptr_t)src & 3) == 0) return *reinterpret_cast<const unsigned *>(src);
        unsigned w = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) if (k < nvalid) w |= (unsigned)(uint8_t)src[k] << (8 * k);
        return w;
    };
    // 16 bytes per lane when the run is inside the block and 16-byte aligned (ld a multiple of 16: every bench shape), else four guarded dwords
    auto load16 = [&](const int8_t *src, int nvalid) -> uint4 {
        if (nvalid >= 16 && ((uintptr_t)src & 15) == 0) return *reinterpret_cast<const uint4 *>(src);
        return make_uint4(load4(src, nvalid), load4(src + 4, nvalid - 4), load4(src + 8, nvalid - 8), load4(src + 12, nvalid - 12));
    };
    if (!TR) {
        // rows u (255) x 128 bytes: thread = (row-in-pass tid >> 3, 16-byte piece tid & 7)
        const int x = tid & 7;
        for (int u = tid >> 3; u < SK_U; u += 32) {
            int i = S0 + J0 + u; if (i >= n) i -= n;
            const int j = J0 + 16 * x;
            const uint4 w = i < r ? load16(blk + (size_t)i * ld + j, c - j) : make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4 *>(tile + u * SK_PITCH + 16 * x) = w;
        }
    } else {
        // element (i, j) lives at blk[j*ld + i]: rows jj (128) x 256 contiguous bytes along u (S0 + J0 and the wrap
        // point are multiples of 128, so a 16-byte piece never straddles the wrap); kept in that orientation
        const int x = tid & 15;
        for (int jj = tid >> 4; jj < SK_T; jj += 16) {
            const int u = 16 * x;
            int i = S0 + J0 + u; if (i >= n) i -= n;
            const int j = J0 + jj;
            const uint4 w = j < c ? load16(blk + (size_t)j * ld + i, r - i) : make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4 *>(tile + jj * SK_PITCH_T + u) = w;
        }
    }
    __syncthreads();
    // thread: columns 4q .. 4q+3, shifts s0 .. s0+3 for four s0.  D[S0+s][J0+jj] = X-tile element (u = s + jj, jj)
    const int q = tid & 31;
#pragma unroll
    for (int it = 0; it < 4; it++) {
        const int s0 = 4 * ((tid >> 5) + 8 * it);
        unsigned out[4];
        if (!TR) {
            unsigned W[7], E[6];
#pragma unroll
            for (int k = 0; k < 7; k++) W[k] = *reinterpret_cast<const unsigned *>(tile + (s0 + 4 * q + k) * SK_PITCH + 4 * q);
#pragma unroll
            for (int k = 0; k < 6; k++) E[k] = __builtin_amdgcn_perm(W[k + 1], W[k], 0x07020500u);          // [W_k.b0, W_k+1.b1, W_k.b2, W_k+1.b3]
#pragma unroll
            for (int m = 0; m < 4; m++) out[m] = __builtin_amdgcn_perm(E[m + 2], E[m], 0x07060100u);        // [W_m.b0, W_m+1.b1, W_m+2.b2, W_m+3.b3]
        } else {
            unsigned V[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {                                    // row jj = 4q + t, bytes s0 + 4q + t .. + 3: shift t inside an aligned pair
                const unsigned *rowp = reinterpret_cast<const unsigned *>(tile + (4 * q + t) * SK_PITCH_T + s0 + 4 * q);
                V[t] = t ? __builtin_amdgcn_alignbyte(rowp[1], rowp[0], (unsigned)t) : rowp[0];
            }
            const unsigned P0 = __builtin_amdgcn_perm(V[1], V[0], 0x05010400u), P1 = __builtin_amdgcn_perm(V[1], V[0], 0x07030602u);
            const unsigned Q0 = __builtin_amdgcn_perm(V[3], V[2], 0x05010400u), Q1 = __builtin_amdgcn_perm(V[3], V[2], 0x07030602u);
            out[0] = __builtin_amdgcn_perm(Q0, P0, 0x05040100u); out[1] = __builtin_amdgcn_perm(Q0, P0, 0x07060302u);
            out[2] = __builtin_amdgcn_perm(Q1, P1, 0x05040100u); out[3] = __builtin_amdgcn_perm(Q1, P1, 0x07060302u);
        }
#pragma unroll
        for (int m = 0; m < 4; m++) *reinterpret_cast<unsigned *>(D + (size_t)(S0 + s0 + m) * n + J0 + 4 * q) = sk_clean(out[m], square);
    }
}

// ---------------------------------------------------------------- FFT encode
// Exchange images are indexed through an XOR swizzle instead of padding: address bits 0..4 (the 32 eight-byte bank pairs of a 256-byte bank
// sweep) are XORed with index bits 3..7.  Every access pattern of the kernel - 64 lanes that vary any six of the index bits 0..7 with the others
// fixed (contiguous, stride 4, stride 32, bit-reversed) - then maps onto all 32 bank pairs exactly twice, the minimum for 512 bytes.
__device__ __forceinline__ int padj(int j) { return j ^ ((j >> 3) & 31); }
// the last exchange (bit-reversed writers: 32 lanes vary index bits 4..8; readers take consecutive words) uses index bits 5..8 instead
__device__ __forceinline__ int padj_fin(int j) { return j ^ ((j >> 5) & 15); }

// (ar + i ai) * (wr + i wi) with each component as ONE double-double dot product (two products share the final
// renormalisation): 19 flops per component instead of 2 dd_mul + 1 dd_add = 25.
__device__ __forceinline__ dd dd_dot2(dd a, dd w, dd b, dd x, double sgn) {          // a*w + sgn*b*x, sgn = +-1
    const double p1 = a.hi * w.hi, e1 = fma(a.hi, w.hi, -p1);
    const double bh = sgn * b.hi, bl = sgn * b.lo;
    const double p2 = bh * x.hi, e2 = fma(bh, x.hi, -p2);
    dd s = dd_two_sum(p1, p2);
    double lo = e1 + e2;
    lo = fma(a.hi, w.lo, lo); lo = fma(a.lo, w.hi, lo);
    lo = fma(bh, x.lo, lo); lo = fma(bl, x.hi, lo);
    return dd_quick(s.hi, s.lo + lo);
}
// the same without the final renormalisation: (hi, lo) with |lo| a few ulps of hi - all that dd_round_away needs
__device__ __forceinline__ dd dd_dot2_raw(dd a, dd w, dd b, dd x, double sgn) {
    const double p1 = a.hi * w.hi, e1 = fma(a.hi, w.hi, -p1);
    const double bh = sgn * b.hi, bl = sgn * b.lo;
    const double p2 = bh * x.hi, e2 = fma(bh, x.hi, -p2);
    dd s = dd_two_sum(p1, p2);
    double lo = e1 + e2;
    lo = fma(a.hi, w.lo, lo); lo = fma(a.lo, w.hi, lo);
    lo = fma(bh, x.lo, lo); lo = fma(bl, x.hi, lo);
    return dd_make(s.hi, s.lo + lo);
}
__device__ __forceinline__ void cdd_mul_ip(dd &re, dd &im, dd wr, dd wi) {
    const dd r = dd_dot2(re, wr, im, wi, -1.0), i = dd_dot2(re, wi, im, wr, 1.0);
    re = r; im = i;
}
// ---- fixed-grid double-double for the FFT of GENOTYPE rows.  Every intermediate of that transform is bounded by sum |z_m| <= 4096 * |128 + 128i| < 2^20
// for ANY int8 row (genotypes after missing -> 0 and squaring are <= 4: < 2^15), so the high parts can live on the fixed grid 2^-31 Z (|hi| < 2^20:
// 51 bits): two grid numbers add EXACTLY in one plain addition, the low parts (|lo| <= 2^-32 after a product, <= 2^-29 after the three add levels of
// a radix-8 pass) in another - 2 flops per sum instead of 8.  Only products leave the grid; they are put back by the magic-number split
// hi' = (p + M) - M, lo' = (p - hi') + e with M = 1.5 * 2^21 (4 flops, which replace the 3 of the renormalisation they had).  Absolute error: low-part
// sums 2^-83 each, products 2^-85: ~2^-80 after four passes, ~2^-59 on a scaled coefficient - far inside the 2^-40 band of the near-tie audit.
// Arbitrary real slot vectors (F64IN) are unbounded and keep the general path.
constexpr double GRID_M = 3145728.0;                      // 1.5 * 2^21: ulp(M) = 2^-31
__device__ __forceinline__ dd grid_split(double p, double e) { const double h = (p + GRID_M) - GRID_M; return dd_make(h, (p - h) + e); }
template <bool GRID> __device__ __forceinline__ dd fx_add(dd a, dd b) { return GRID ? dd_make(a.hi + b.hi, a.lo + b.lo) : dd_add_lazy(a, b); }
template <bool GRID> __device__ __forceinline__ dd fx_sub(dd a, dd b) { return GRID ? dd_make(a.hi - b.hi, a.lo - b.lo) : dd_sub_lazy(a, b); }
template <bool GRID> __device__ __forceinline__ dd fx_mul(dd a, dd b) {          // dd_mul, result on the grid
    double p = a.hi * b.hi, e = fma(a.hi, b.hi, -p);
    e = fma(a.hi, b.lo, e); e = fma(a.lo, b.hi, e);
    return GRID ? grid_split(p, e) : dd_quick(p, e);
}
template <bool GRID> __device__ __forceinline__ dd fx_dot2(dd a, dd w, dd b, dd x, double sgn) {   // dd_dot2, result on the grid
    const double p1 = a.hi * w.hi, e1 = fma(a.hi, w.hi, -p1);
    const double bh = sgn * b.hi, bl = sgn * b.lo;
    const double p2 = bh * x.hi, e2 = fma(bh, x.hi, -p2);
    if (GRID) {
        // each product is split on the grid by itself (the remainders p - h are exact, |.| <= 2^-32), the grid parts add exactly: no two_sum
        const double h1 = (p1 + GRID_M) - GRID_M, h2 = (p2 + GRID_M) - GRID_M;
        double lo = (p1 - h1) + (p2 - h2);
        lo += e1 + e2;
        lo = fma(a.hi, w.lo, lo); lo = fma(a.lo, w.hi, lo);
        lo = fma(bh, x.lo, lo); lo = fma(bl, x.hi, lo);
        return dd_make(h1 + h2, lo);
    }
    dd s = dd_two_sum(p1, p2);
    double lo = e1 + e2;
    lo = fma(a.hi, w.lo, lo); lo = fma(a.lo, w.hi, lo);
    lo = fma(bh, x.lo, lo); lo = fma(bl, x.hi, lo);
    return dd_quick(s.hi, s.lo + lo);
}
#ifndef SFG_ENC_DIAG
#define SFG_ENC_DIAG 0          // timing diagnostics only (wrong results): 1 no pass-twiddle loads, 2 no recombination arithmetic, 4 no exchanges, 8 no twiddle products, 16 no recombination-twiddle loads
#endif
__device__ __forceinline__ void tw_at(const double4 *tab, int idx, dd &wr, dd &wi) {
    if (SFG_ENC_DIAG & 1) { wr = dd_make(0.7 + idx * 1e-9, 1e-18); wi = dd_make(0.3 - idx * 1e-9, 2e-18); return; }
    const double4 w = tab[idx];
    wr = dd_make(w.x, w.y); wi = dd_make(w.z, w.w);
}
// One radix-8 DIF pass on 8 register-resident points at stride S of a sub-transform of length 8S: identical to three
// radix-2 DIF stages (pairs (i,i+4), (i,i+2), (i,i+1)) with the twiddles regrouped - the 12 twiddle products of the
// radix-2 form become 7 output products W^(e t), e = bitrev(r), plus two rotations by 1/8 turn; t = j mod S.
template <int S, bool GRID>
__device__ __forceinline__ void dif_radix8(dd (&xr)[8], dd (&xi)[8], const double4 *tab, int t) {          // tab: this pass's [7][S] twiddle list
    const dd rs = dd_make(7.071067811865475727e-01, -4.833646656726456726e-17);      // 1/sqrt(2) in double-double
    dd ur[4], ui[4], dr[4], di[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        ur[i] = fx_add<GRID>(xr[i], xr[i + 4]); ui[i] = fx_add<GRID>(xi[i], xi[i + 4]);
        dr[i] = fx_sub<GRID>(xr[i], xr[i + 4]); di[i] = fx_sub<GRID>(xi[i], xi[i + 4]);
    }
    {   // d1 *= W8 = (1 - i)/sqrt2: (a + bi) -> ((a + b) + (b - a) i)/sqrt2
        dd a = dr[1], b = di[1];
        dr[1] = fx_mul<GRID>(fx_add<GRID>(a, b), rs); di[1] = fx_mul<GRID>(fx_sub<GRID>(b, a), rs);
    }
    {   // d2 *= -i: (a + bi) -> (b - ai)
        dd a = dr[2]; dr[2] = di[2]; di[2] = dd_neg(a);
    }
    {   // d3 *= W8^3 = (-1 - i)/sqrt2: (a + bi) -> ((b - a) - (a + b) i)/sqrt2
        dd a = dr[3], b = di[3];
        dr[3] = fx_mul<GRID>(fx_sub<GRID>(b, a), rs); di[3] = dd_neg(fx_mul<GRID>(fx_add<GRID>(a, b), rs));
    }
    auto quad = [&](dd (&hr)[4], dd (&hi)[4], int o) {
        dd p0r = fx_add<GRID>(hr[0], hr[2]), p0i = fx_add<GRID>(hi[0], hi[2]);
        dd p1r = fx_add<GRID>(hr[1], hr[3]), p1i = fx_add<GRID>(hi[1], hi[3]);
        dd q0r = fx_sub<GRID>(hr[0], hr[2]), q0i = fx_sub<GRID>(hi[0], hi[2]);
        dd t1r = fx_sub<GRID>(hr[1], hr[3]), t1i = fx_sub<GRID>(hi[1], hi[3]);
        dd q1r = t1i, q1i = dd_neg(t1r);                                               // * -i
        xr[o + 0] = fx_add<GRID>(p0r, p1r); xi[o + 0] = fx_add<GRID>(p0i, p1i);
        xr[o + 1] = fx_sub<GRID>(p0r, p1r); xi[o + 1] = fx_sub<GRID>(p0i, p1i);
        xr[o + 2] = fx_add<GRID>(q0r, q1r); xi[o + 2] = fx_add<GRID>(q0i, q1i);
        xr[o + 3] = fx_sub<GRID>(q0r, q1r); xi[o + 3] = fx_sub<GRID>(q0i, q1i);
    };
    quad(ur, ui, 0);
    quad(dr, di, 4);
    if (S > 1) {                                                                       // S == 1: t = 0, every output twiddle is 1
#pragma unroll
        for (int r = 1; r < 8; r++) {                                                  // output r carries W_{8S}^(bitrev3(r) t)
            if (SFG_ENC_DIAG & 8) continue;
            dd wr, wi; tw_at(tab, (r - 1) * S + t, wr, wi);
            const dd pr = fx_dot2<GRID>(xr[r], wr, xi[r], wi, -1.0), pi = fx_dot2<GRID>(xr[r], wi, xi[r], wr, 1.0);
            xr[r] = pr; xi[r] = pi;
        }
    }
}

// The double-double pipeline carries ~2^-100 relative error (absolute ~2^-65 on these magnitudes; ~2^-59 on the fixed grid), the reference's EncoderBig 256
// bits.  A coefficient whose exact value lies within 2^-40 of a rounding tie is counted (sfg_ctx_encoder_near_ties): the two
// encoders can only disagree on such a coefficient, so a zero count PROVES the block was rounded as the reference rounds it.
// A second, sticky counter takes the coefficients within 2^-50 of a tie (2^8 times the pipeline's error estimate, about once per 10^15
// coefficients): while it is non-zero every synchronising entry point FAILS (sfg_encoder_check) - the contract is bit-exactness, and such a
// coefficient has to be re-derived by a big-float encoder on the host before the product may be used.
template <bool EXACT_TIES>
__device__ __forceinline__ double dd_round_away(dd x, unsigned &near_tie, double band, bool &inside) {                // integer-valued double
    double nn = __builtin_rint(x.hi);
    double diff = (x.hi - nn) + x.lo;                                                         // |diff| <= 1/2 + |x.lo|
    const double tie_dist = __builtin_fabs(__builtin_fabs(diff) - 0.5);
    inside = tie_dist < band;                                                                 // band (2^-50): the double-double value does not prove the rounding - re-derived exactly below, or the call FAILS
    near_tie += (tie_dist < 0x1p-40 ? 1u : 0u);                                               // audit band
    if (!EXACT_TIES) return nn + __builtin_rint(diff);      // = the rule below whenever |diff| != 1/2; an exact tie (impossible for integer slot values at Delta/n = 2^k) is counted above
    const bool up = (diff > 0.5) | ((diff == 0.5) & (nn >= 0)), dn = (diff < -0.5) | ((diff == -0.5) & (nn <= 0));      // (no short-circuit: selects, not branches)
    return nn + (up ? 1.0 : 0.0) - (dn ? 1.0 : 0.0);
}

// Exact re-derivation of ONE coefficient of a genotype-row plaintext (Delta / n = 2^sexp).  p_j = (Delta / n) sum_t v_t cos(2 pi 5^t j / 2N): with 5^t = 4 m + 1
// over the FFT's input order m every term is a small integer times a table cosine; the cosines enter as 100-bit fixed point (three signed limbs of 40 bits from the
// {hi, lo} pair, exact), the 8192 products are summed as integers (no rounding at all), and the only error left is the table's: below 2^-68 of a unit after the
// scaling.  A sum farther than 2^-62 from the tie PROVES its rounding.  Called by the one lane that met a coefficient inside the band (about once per 10^15
// coefficients): a serial loop, deliberately not inlined so that it costs the encoder's hot path one predicate and no registers.
__device__ __noinline__ bool enc_tie_resolve(const int8_t *row, const uint16_t *tinv, int nrot, const double2 *costab, int sexp, int j, double *out) {
    const int n = SFG_SLOTS;
    __int128 sa = 0, sb = 0, sc = 0;                                    // (the matrix API takes any int8: |v| 128 x |limb| 2^41 x 8192 terms passes 2^63; the loop is cold)
    for (int m = 0; m < n; m++) {
        int t = (int)tinv[m] - nrot; t += t < 0 ? n : 0;
        const int v = (int)row[t];                                      // (the skewed block: missing calls are zero already)
        if (!v) continue;
        unsigned k = ((4u * (unsigned)m + 1u) * (unsigned)j) & 32767u;  // angle 2 pi k / 32768
        if (k > 16384u) k = 32768u - k;                                 // cos is even
        const bool neg = k > 8192u; if (neg) k = 16384u - k;            // cos(pi - x) = -cos x
        const double2 cv = costab[k];
        const double x = cv.x * 0x1p20, a = __builtin_rint(x), r1 = x - a;
        const double y = r1 * 0x1p40, bq = __builtin_rint(y), r2 = y - bq;
        const double cq = __builtin_rint(r2 * 0x1p40 + cv.y * 0x1p100);
        const long long sv = neg ? -(long long)v : (long long)v;
        sa += (__int128)(sv * (long long)a); sb += (__int128)(sv * (long long)bq); sc += (__int128)(sv * (long long)cq);     // each product < 2^7 2^41: exact in 64 bits
    }
    const __int128 T = (sa << 80) + (sb << 40) + sc;      // p_j 2^(100 - sexp)
    const int sh = 100 - sexp;
    const __int128 q = T >> sh, rem = T - (q << sh), half = (__int128)1 << (sh - 1);      // floor; 0 <= rem < 2^sh
    const __int128 dist = rem > half ? rem - half : half - rem;
    if (dist <= ((__int128)1 << (sh - 62))) return false;
    *out = (double)((long long)q + (rem > half ? 1 : 0));
    return true;
}

// rows: diag-major int8 rows of length n; plaintext p encodes row (shift0 + p) right-rotated by d*((shift0+p)/d).
// F64IN: rows are n doubles (arbitrary real slot vectors, no rotation) — the Mask / EncodeFloatVector use.
constexpr size_t ENC_LDS_BYTES = (size_t)2 * ENC_H * 8;        // 65,536 B: two workgroups per CU
// Every exchange moves the HIGH parts of all 4096 points through the 64 KiB image (re, im: 2 x 4096 doubles), then the LOW parts: the image
// holds half of the double-double data at a time, every thread does the same work in both rounds (no divergent writers), and a thread carries
// at most 8 high + 8 low complex parts across a round.  16 waves per CU.
template <bool F64IN>
__global__ void __launch_bounds__(512, 4) k_fft_encode(const void *Dv, int shift0, const double4 *tb, const uint16_t *tinv,
                                                      double *pc_out, unsigned long long *tie_count, const double2 *costab, int sexp, double band) {
    unsigned near_tie = 0;
    extern __shared__ double lds[];
    double *RE = lds, *IM = lds + ENC_H;
    const int n = SFG_SLOTS, h = ENC_H, tid = threadIdx.x;
    const int shift = shift0 + blockIdx.x;
    const int nrot = F64IN ? 0 : SFG_D * (shift / SFG_D);          // matmult.go:1426: nrot = d * giant
    const int8_t *row = (const int8_t *)Dv + (size_t)shift * n;
    const double *rowd = (const double *)Dv + (size_t)shift * n;
    // stage the 8 KiB row at the start of LDS (the image is first written after pass 1)
    int8_t *rowl = reinterpret_cast<int8_t *>(lds);
    if (!F64IN) reinterpret_cast<uint4 *>(rowl)[tid] = reinterpret_cast<const uint4 *>(row)[tid];
    __syncthreads();
    dd xr[8], xi[8], yr[8], yi[8];
    // src[k] goes to image index widx(k); dst[k] comes from index ridx(k).  WAVE: every index a wave writes or reads lies in its own 512-point
    // region (exchanges 2 -> 3 and 3 -> 4 are transposes inside a wave), so after one workgroup barrier (the previous exchange's readers) only
    // wave-level ordering is needed and the waves drift freely.
    auto exchange = [&](dd (&sr)[8], dd (&si)[8], dd (&dr)[8], dd (&di)[8], auto widx, auto ridx, auto wave, bool entry_barrier) {
        constexpr bool WAVE = decltype(wave)::value;
        auto sync = [&]() { if (WAVE) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); else __syncthreads(); };
        if (SFG_ENC_DIAG & 4) { for (int k = 0; k < 8; k++) { dr[k] = sr[k]; di[k] = si[k]; } return; }
        if (entry_barrier) __syncthreads(); else sync();             // the row staging area / the previous readers are done
#pragma unroll
        for (int k = 0; k < 8; k++) { const int p = padj(widx(k)); RE[p] = sr[k].hi; IM[p] = si[k].hi; }
        sync();
#pragma unroll
        for (int k = 0; k < 8; k++) { const int p = padj(ridx(k)); dr[k].hi = RE[p]; di[k].hi = IM[p]; }
        sync();
#pragma unroll
        for (int k = 0; k < 8; k++) { const int p = padj(widx(k)); RE[p] = sr[k].lo; IM[p] = si[k].lo; }
        sync();
#pragma unroll
        for (int k = 0; k < 8; k++) { const int p = padj(ridx(k)); dr[k].lo = RE[p]; di[k].lo = IM[p]; }
    };
    const std::false_type wg_wide; const std::true_type wave_local;
    // ---- pass 1: bits a; thread = (b,c,d) = tid, element j = a*512 + tid
#pragma unroll
    for (int a = 0; a < 8; a++) {
        const int m = a * 512 + tid;                               // z_m = u_2m + i u_2m+1, u_mm = v[tinv[mm]] = row[(tinv[mm] - nrot) mod n]
        int t0 = (int)tinv[2 * m] - nrot, t1 = (int)tinv[2 * m + 1] - nrot;
        t0 += t0 < 0 ? n : 0; t1 += t1 < 0 ? n : 0;
        if (F64IN) { xr[a] = dd_make(rowd[t0], 0.0); xi[a] = dd_make(rowd[t1], 0.0); }
        else { xr[a] = dd_make((double)rowl[t0], 0.0); xi[a] = dd_make((double)rowl[t1], 0.0); }
    }
    dif_radix8<512, !F64IN>(xr, xi, tb + ENC_TB_P512, tid);
    {
        // ---- exchange 1 -> 2.  Writer tid = (b, cd) holds a = 0..7; reader (a, cd) needs b = 0..7 of j = a*512 + b*64 + cd
        const int cd = tid & 63, ar = tid >> 6;
        exchange(xr, xi, yr, yi, [&](int a) { return a * 512 + tid; }, [&](int b) { return ar * 512 + b * 64 + cd; }, wg_wide, true);
        dif_radix8<64, !F64IN>(yr, yi, tb + ENC_TB_P64, cd);
        // ---- exchange 2 -> 3.  Writer (a, c, d) holds b = 0..7; reader (ab, d) needs c = 0..7 of j = ab*64 + c*8 + d
        const int d = tid & 7, ab = tid >> 3;
        exchange(yr, yi, xr, xi, [&](int b) { return ar * 512 + b * 64 + cd; }, [&](int c) { return ab * 64 + c * 8 + d; }, wave_local, true);
        dif_radix8<8, !F64IN>(xr, xi, tb + ENC_TB_P8, d);
        // ---- exchange 3 -> 4.  Writer (ab, d) holds c = 0..7; reader tid = abc needs d = 0..7 of j = tid*8 + d
        exchange(xr, xi, yr, yi, [&](int c) { return ab * 64 + c * 8 + d; }, [&](int d4) { return tid * 8 + d4; }, wave_local, false);
        dif_radix8<1, !F64IN>(yr, yi, tb, 0);
    }
    // ---- recombination.  Position p = tid*8 + d holds Z_c with c = brev12(p) = brev3(d) << 9 | brev9(tid): the results are stored under c, and
    // thread tid takes the pairs (c, h - c), c = tid + 512 i, i = 0..3 (and thread 0 the self-paired c = h/2): consecutive lanes read consecutive
    // image words and twiddle entries and write consecutive coefficients.
    double *pc = pc_out + (size_t)blockIdx.x * n;
    const int cbase = (int)(__brev((unsigned)tid) >> 23);
    dd Ar[5], Ai[5], Br[5], Bi[5];
    const int npair = tid == 0 ? 5 : 4;
    __syncthreads();
#pragma unroll
    for (int d4 = 0; d4 < 8; d4++) { const int p = padj_fin((int)((__brev((unsigned)d4) >> 29) << 9) | cbase); RE[p] = yr[d4].hi; IM[p] = yi[d4].hi; }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 5; i++) if (i < npair) {
        const int c = tid + 512 * i, pa = padj_fin(c), pb = padj_fin((h - c) & (h - 1));
        Ar[i].hi = RE[pa]; Ai[i].hi = IM[pa]; Br[i].hi = RE[pb]; Bi[i].hi = -IM[pb];
    }
    __syncthreads();
#pragma unroll
    for (int d4 = 0; d4 < 8; d4++) { const int p = padj_fin((int)((__brev((unsigned)d4) >> 29) << 9) | cbase); RE[p] = yr[d4].lo; IM[p] = yi[d4].lo; }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 5; i++) if (i < npair) {
        const int c = tid + 512 * i, pa = padj_fin(c), pb = padj_fin((h - c) & (h - 1));
        Ar[i].lo = RE[pa]; Ai[i].lo = IM[pa]; Br[i].lo = RE[pb]; Bi[i].lo = -IM[pb];
    }
    // a coefficient inside the band is remembered (one per lane: the band holds ~10^-15 of them) and re-derived exactly at the very end of the kernel, where nothing
    // else is live: the hot path pays two selects
    int tie_j = -1; unsigned tie_n = 0;
    auto note = [&](int j) { tie_j = j; tie_n++; };
    // one pair (c, h - c): A = Z_c, B = conj Z_{h-c};  wo = omega^-c / 2 = zeta^-4c / 2, zc = (Delta/n) zeta^-c, zh = (Delta/n) zeta^-(h-c)
    auto recomb = [&](int c, dd Ar, dd Ai, dd Br, dd Bi) {
        // (genotype rows: A and B are sums on the fixed grid, so the recombination adds are exact two-flop grid adds as well)
        constexpr bool GRID = !F64IN;
        auto radd = [](dd a, dd b) { return GRID ? dd_make(a.hi + b.hi, a.lo + b.lo) : dd_add(a, b); };
        auto rsub = [](dd a, dd b) { return GRID ? dd_make(a.hi - b.hi, a.lo - b.lo) : dd_sub(a, b); };
        auto half = [](dd a) { return dd_make(a.hi * 0.5, a.lo * 0.5); };                         // exact
        if (SFG_ENC_DIAG & 2) { pc[c] = Ar.hi; if (c > 0) pc[n - c] = Ai.hi; if (c < h / 2) { pc[h - c] = Br.hi; pc[n - (h - c)] = Bi.hi; } return; }
        const bool nt = SFG_ENC_DIAG & 16;
        const double4 wo = nt ? make_double4(0.7 + c * 1e-9, 1e-18, 0.3, 1e-18) : tb[ENC_TB_RW + c];
        const double4 zc = nt ? make_double4(0.6 + c * 1e-9, 1e-18, 0.4, 1e-18) : tb[ENC_TB_RZ + c];
        dd Xr = half(radd(Ar, Br)), Xi = half(radd(Ai, Bi));
        dd Dr = rsub(Ar, Br), Di = rsub(Ai, Bi);                     // (the table holds omega^-c / 2: exact halving, one product instead of four)
        dd Or = Di, Oi = dd_neg(Dr);                                // (A-B)/i = -i (A-B)
        dd wor = dd_make(wo.x, wo.y), woi = dd_make(wo.z, wo.w);
        dd Yr = fx_dot2<GRID>(Or, wor, Oi, woi, -1.0), Yi = fx_dot2<GRID>(Or, woi, Oi, wor, 1.0);
        // W_c
        {
            dd Wr = radd(Xr, Yr), Wi = radd(Xi, Yi);
            dd zr = dd_make(zc.x, zc.y), zi = dd_make(zc.z, zc.w);
            dd wr = dd_dot2_raw(Wr, zr, Wi, zi, -1.0), wi = dd_dot2_raw(Wr, zi, Wi, zr, 1.0);       // (rounded next: no renormalisation needed)
            bool in0, in1 = false;
            pc[c] = dd_round_away<F64IN>(wr, near_tie, band, in0);
            if (c > 0) pc[n - c] = -dd_round_away<F64IN>(wi, near_tie, band, in1);
            if (in0) note(c);
            if (in1) note(n - c);
        }
        // W_{h-c}  (c = 0 gives W_h)
        if (c < h / 2) {
            const int cc = h - c;
            const double4 zh = nt ? make_double4(0.5 + c * 1e-9, 1e-18, 0.45, 1e-18) : tb[ENC_TB_RZ2 + c];
            dd Wr = rsub(Xr, Yr), Wi = dd_neg(rsub(Xi, Yi));
            dd zr = dd_make(zh.x, zh.y), zi = dd_make(zh.z, zh.w);
            dd wr = dd_dot2_raw(Wr, zr, Wi, zi, -1.0), wi = dd_dot2_raw(Wr, zi, Wi, zr, 1.0);       // (rounded next: no renormalisation needed)
            bool in0, in1 = false;
            pc[cc] = dd_round_away<F64IN>(wr, near_tie, band, in0);
            if (cc < h) pc[n - cc] = -dd_round_away<F64IN>(wi, near_tie, band, in1);
            if (in0) note(cc);
            if (in1) note(n - cc);
        }
    };
#pragma unroll
    for (int i = 0; i < 4; i++) recomb(tid + 512 * i, Ar[i], Ai[i], Br[i], Bi[i]);
    if (tid == 0) recomb(h / 2, Ar[4], Ai[4], Br[4], Bi[4]);
    if (tie_n) {       // re-derived exactly by this lane (enc_tie_resolve), or counted as unproven (a second one in the same lane, real-valued slot rows, Delta / n not a
        unsigned unproven = tie_n - 1, resolved = 0;      // power of two, the A/B build -DSFG_ENC_NO_RESOLVE) - which makes the synchronising entry points fail
        double v;
#ifndef SFG_ENC_NO_RESOLVE
        if (!F64IN && sexp >= 0 && enc_tie_resolve(row, tinv, nrot, costab, sexp, tie_j, &v)) { pc[tie_j] = v; resolved = 1; } else
#endif
            unproven++;
        if (resolved) atomicAdd(tie_count + 2, 1ULL);
        if (unproven) atomicAdd(tie_count + 1, (unsigned long long)unproven);
    }
    if (near_tie) atomicAdd(tie_count, (unsigned long long)near_tie);
}

static int enc_pc_scratch(sfg_ctx *ctx, size_t nplain, double **pc) {
    return sfg_scratch(ctx, "enc.pc", nplain * SFG_SLOTS * sizeof(double), (void **)pc);
}
int encode_set_attrs(sfg_ctx *ctx) {
    SFG_HIP(ctx, hipFuncSetAttribute((const void *)k_fft_encode<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ENC_LDS_BYTES));
    SFG_HIP(ctx, hipFuncSetAttribute((const void *)k_fft_encode<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ENC_LDS_BYTES));
    return 0;
}

int launch_skew(sfg_ctx *ctx, const int8_t *blk, size_t ld, int r, int c, int transposed, int square, int8_t *D) {
    if (transposed) hipLaunchKernelGGL(k_skew<true>, dim3(SFG_SLOTS / SK_T, SFG_SLOTS / SK_T), dim3(256), 0, ctx->stream, blk, ld, r, c, square, D);
    else hipLaunchKernelGGL(k_skew<false>, dim3(SFG_SLOTS / SK_T, SFG_SLOTS / SK_T), dim3(256), 0, ctx->stream, blk, ld, r, c, square, D);
    SFG_HIP(ctx, hipGetLastError());
    return 0;
}

// encode diagonals [shift0, shift0+nshift) of a skewed block D into pt[nshift][L][N]
// half_rows + G > 0: `pt` is the base of a grouped panel and rows are scattered by PanelMap (half rows only)
int encode_rows_launches(const sfg_ctx *ctx, int nshift) { const int B = ctx->cfg.enc_batch; return (nshift + B - 1) / B; }
int launch_encode_rows(sfg_ctx *ctx, const int8_t *D, int shift0, int nshift, int L, u64 *pt, bool half_rows, int G, int g, unsigned packed_mask, const PcCache *pcache, StagePack *sp, PtRide *ride) {
    EncTables *et = (EncTables *)ctx->enc_tables();
    const size_t lds_bytes = ENC_LDS_BYTES;
    if (sp && (shift0 % SFG_D || !half_rows || !(packed_mask >> 31))) SFG_FAIL(ctx, "encode: internal: the streamed transposition takes whole giant steps of digit-plane rows");
    const int BATCH = sp ? ctx->cfg.stage_giants * SFG_D : ctx->cfg.enc_batch;                 // plaintexts per FFT / NTT launch pair (SFG_ENC_BATCH)
    const int cmode = pcache && pcache->slot && half_rows && G > 0 ? pcache->mode : 0;
    double *pc = nullptr;
    if (!cmode) SFG_TRY(enc_pc_scratch(ctx, (size_t)(nshift < BATCH ? nshift : BATCH), &pc));
    for (int s0 = 0; s0 < nshift; s0 += BATCH) {
        const int nb = nshift - s0 < BATCH ? nshift - s0 : BATCH;
        if (cmode == 1) pc = pcache->slot + (size_t)(shift0 + s0) * SFG_SLOTS;          // the coefficient rows of these shifts live in the cache slot
        if (cmode <= 1) {
            PhaseTimer t(ctx, "encode", false);
            hipLaunchKernelGGL(k_fft_encode<false>, dim3(nb), dim3(512), lds_bytes, ctx->stream, (const void *)D, shift0 + s0, et->tb, et->tinv, pc, (unsigned long long *)ctx->tie_count_dev, et->costab, et->sexp, ctx->cfg.tie_band);
            SFG_HIP(ctx, hipGetLastError());
        }
        if (cmode == 2) pc = pcache->slot + (size_t)(shift0 + s0) * SFG_SLOTS;
        // the panel NTT is timed on a sample (every 16th launch carries an event pair: 50 000 launches per power iteration) and counted in full
        // (launches that carry mover workgroups of the riding transposition are timed and counted apart: phases ntt_plain_ride / ntt_ride_all, their mover bytes in pt_ride)
        const bool sampled = half_rows && G > 0 && (ctx->ntt_plain_seq++ & 15) == 0;
        const bool riding = !sp && half_rows && G > 0 && ride && ride->on && ride->next < ride->total();
        PhaseTimer tn(ctx, riding ? "ntt_plain_ride" : "ntt_plain", sampled);
        if (half_rows && G > 0) { PhaseStat &all = ctx->phases[riding ? "ntt_ride_all" : "ntt_plain_all"]; all.launches += 1; }
        if (sp) {
            // the staging buffer is free once the previous batch has been transposed (that ran beside this batch's FFT)
            if (sp->pending) SFG_HIP(ctx, hipStreamWaitEvent(ctx->stream, sp->ev_pack, 0));
            PanelMap pm{0, 0, shift0 + s0, packed_mask};
            SFG_TRY(launch_ntt_plain_half(ctx, cmode == 3 ? pcache->slot : pc, sp->stage, nb, L, pm, cmode == 3 ? pcache->perm : nullptr));
            SFG_HIP(ctx, hipEventRecord(sp->ev_ntt, ctx->stream));
            SFG_HIP(ctx, hipStreamWaitEvent(sp->q, sp->ev_ntt, 0));
            SFG_TRY(launch_i8_pack_stage(ctx, *sp, shift0 + s0, nb, L));
            SFG_HIP(ctx, hipEventRecord(sp->ev_pack, sp->q));
            sp->pending = true;
        }
        else if (half_rows && G > 0) {
            // the riding transposition: this launch's share of the previous MAC launch's panel goes along as mover workgroups (k_ntt_half3_move)
            MoveJob mj; const MoveJob *mv = nullptr; double moved = 0;
            if (ride && ride->on && ride->next < ride->total()) {
                mj = ride->job; mj.first = ride->next; mj.count = std::min(ride->per, ride->total() - ride->next); ride->next += mj.count; mv = &mj;
                const unsigned hi = mj.first + mj.count, n5 = mj.n5, in5 = mj.first < n5 ? std::min(hi, n5) - mj.first : 0u;
                moved = in5 * ride->item_bytes5 + (mj.count - in5) * ride->item_bytes6;
                if (sampled) { PhaseStat &ps = ctx->phases["pt_ride"]; ps.launches += 1; ps.bytes += moved; }       // (the timed launches' share: same sample as ntt_plain_ride)
            }
            PanelMap pm{G, g, shift0 + s0, packed_mask};
            if (packed_mask & PT_KMAJOR) pm.K = G * SFG_D;            // K-major panel: a column holds the G block rows' 91 baby steps each
            if (cmode == 3) SFG_TRY(launch_ntt_plain_half(ctx, pcache->slot, pt, nb, L, pm, pcache->perm, mv));
            else SFG_TRY(launch_ntt_plain_half(ctx, pc, pt, nb, L, pm, nullptr, mv));
        }
        else if (half_rows) { PanelMap pm{0, 0, 0}; SFG_TRY(launch_ntt_plain_half(ctx, pc, pt + (size_t)s0 * L * (SFG_N / 2), nb, L, pm)); }
        else SFG_TRY(launch_ntt_plain(ctx, pc, pt + (size_t)s0 * L * SFG_N, nb, L));
        if (sampled) {      // algorithmic bytes: the coefficient row in, L output rows of N/2 words (or five digit planes of N/2 bytes for the int8 MAC's moduli)
            double wr = 0; for (int l = 0; l < L; l++) wr += ((packed_mask >> 31) && ((packed_mask >> l) & 1u)) ? 5.0 : ((packed_mask >> 30) & 1u) ? 6.0 : 8.0;     // (six planes for the 46-bit row of an all-int8 product)
            tn.stop(1, (double)nb * (SFG_N / 2) * (8.0 + wr));
        }
    }
    return 0;
}

extern "C" int sfg_encode_diags_dev(sfg_ctx *ctx, const int8_t *block, size_t ld, int r, int c, int transposed,
                                    int shift0, int nshift, int L, uint64_t *pt) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (r < 1 || c < 1 || r > SFG_SLOTS || c > SFG_SLOTS) SFG_FAIL(ctx, "sfg_encode_diags: block dims out of range");
    if (shift0 < 0 || nshift < 0 || shift0 + nshift > SFG_SLOTS) SFG_FAIL(ctx, "sfg_encode_diags: shift range out of range");
    if (L < 1 || L > ctx->nq) SFG_FAIL(ctx, "sfg_encode_diags: L out of range");
    int8_t *skew = nullptr;
    SFG_TRY(sfg_scratch(ctx, "enc.skew", (size_t)SFG_SLOTS * SFG_SLOTS, (void **)&skew));
    SFG_TRY(launch_skew(ctx, block, ld, r, c, transposed, 0, skew));
    u64 *half = nullptr;
    SFG_TRY(sfg_scratch(ctx, "enc.half", (size_t)nshift * L * (SFG_N / 2) * 8, (void **)&half));
    SFG_TRY(launch_encode_rows(ctx, skew, shift0, nshift, L, half, true));
    return launch_expand_half(ctx, half, (u64 *)pt, (size_t)nshift * L);
}

// EncodeFloatVector-style host helper (crypto.go:398-420 behind Mask/MaskTrunc, basics.go:110-172): real slot
// vectors -> coefficient-domain integers of the scaled inverse embedding (exactly rounded, like EncoderBig).
extern "C" int sfg_encode_coeffs_host(sfg_ctx *ctx, const double *values_host, int nvec, int64_t *coeffs_host) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (nvec <= 0) return 0;
    EncTables *et = (EncTables *)ctx->enc_tables();
    const size_t n = SFG_SLOTS, lds_bytes = ENC_LDS_BYTES;
    double *dv = nullptr; double *dpc = nullptr;
    SFG_HIP(ctx, hipMalloc(&dv, (size_t)nvec * n * 8));
    if (hipMalloc(&dpc, (size_t)nvec * n * 8) != hipSuccess) { (void)hipFree(dv); SFG_FAIL(ctx, "encode_coeffs: out of device memory"); }
    std::vector<double> pc((size_t)nvec * n);
    int rc = 0;
    if (hipMemcpyAsync(dv, values_host, (size_t)nvec * n * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = 1;
    if (!rc) {
        hipLaunchKernelGGL(k_fft_encode<true>, dim3(nvec), dim3(512), lds_bytes, ctx->stream, (const void *)dv, 0, et->tb, et->tinv, dpc, (unsigned long long *)ctx->tie_count_dev, et->costab, -1, ctx->cfg.tie_band);
        if (hipGetLastError() != hipSuccess) rc = 1;
    }
    if (!rc && hipMemcpyAsync(pc.data(), dpc, (size_t)nvec * n * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = 1;
    (void)hipStreamSynchronize(ctx->stream); (void)hipFree(dv); (void)hipFree(dpc);
    if (rc) SFG_FAIL(ctx, "encode_coeffs: device operation failed");
    for (int v = 0; v < nvec; v++) {                       // expand p_n = 0, p_{n+c} = -p_{n-c}
        const double *h = pc.data() + (size_t)v * n; int64_t *o = coeffs_host + (size_t)v * SFG_N;
        for (size_t c = 0; c < n; c++) o[c] = (int64_t)h[c];
        o[n] = 0;
        for (size_t c = 1; c < n; c++) o[n + c] = -(int64_t)h[n - c];
    }
    return 0;
}

// crypto.EncodeFloatVector (crypto.go:398-420 -> encoder.EncodeNTT at `level`, default scale): nvec real slot vectors
// -> NTT-domain plaintexts pt_dev[nvec][level+1][N] (canonical residues).  Feeds Mask / MaskTrunc / CPMult.
extern "C" int sfg_encode_vectors_dev(sfg_ctx *ctx, const double *values_host, int nvec, int level, uint64_t *pt_dev) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (nvec <= 0) return 0;
    if (level < 0 || level >= ctx->nq) SFG_FAIL(ctx, "encode_vectors: level out of range");
    EncTables *et = (EncTables *)ctx->enc_tables();
    const size_t n = SFG_SLOTS, lds_bytes = ENC_LDS_BYTES;
    void *p = nullptr;
    SFG_TRY(sfg_scratch(ctx, "enc.vectors", (size_t)nvec * n * 16, &p));
    double *dv = (double *)p; double *dpc = dv + (size_t)nvec * n;
    SFG_HIP(ctx, hipMemcpyAsync(dv, values_host, (size_t)nvec * n * 8, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_fft_encode<true>, dim3(nvec), dim3(512), lds_bytes, ctx->stream, (const void *)dv, 0, et->tb, et->tinv, dpc, (unsigned long long *)ctx->tie_count_dev, et->costab, -1, ctx->cfg.tie_band);
    SFG_HIP(ctx, hipGetLastError());
    SFG_TRY(launch_ntt_plain(ctx, dpc, (u64 *)pt_dev, (size_t)nvec, level + 1));
    SFG_HIP(ctx, hipStreamSynchronize(ctx->stream));          // values_host may be reused by the caller
    return 0;
}

// encoder coefficients (since context creation / the last reset) whose double-double value lay within 2^-40 of a rounding tie
extern "C" int sfg_ctx_encoder_near_ties(sfg_ctx *ctx, unsigned long long *count, int reset) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    SFG_TRY(sfg_sync_all(ctx));
    unsigned long long c[2] = {0, 0};
    SFG_HIP(ctx, hipMemcpy(c, ctx->tie_count_dev, 16, hipMemcpyDeviceToHost));
    *count = c[0];
    if (reset) {
        if (c[1]) sfg_ptc_invalidate_all(ctx);       // rows cached while an unprovable rounding was outstanding are not served after the reset
        SFG_HIP(ctx, hipMemset(ctx->tie_count_dev, 0, 32));
    }
    return 0;
}

// after a synchronisation of the context's queue: fail while a coefficient too close to a rounding tie to be proven equal to the reference's
// EncoderBig rounding is outstanding (reset with sfg_ctx_encoder_near_ties(ctx, &n, 1) once the affected product has been re-derived)
int sfg_encoder_check(sfg_ctx *ctx) {
    unsigned long long c[2] = {0, 0};
    SFG_HIP(ctx, hipMemcpy(c, ctx->tie_count_dev, 16, hipMemcpyDeviceToHost));
    if (c[1]) sfg_ptc_invalidate_all(ctx);
    if (c[1]) SFG_FAIL(ctx, "encoder: %llu coefficient(s) within 2^-50 of a rounding tie - the double-double encoder cannot prove them rounded as the reference's 256-bit "
                            "EncoderBig rounds; re-derive the products of this context since the last reset with a big-float encoder (sfg_ctx_encoder_near_ties resets)", c[1]);
    return 0;
}
// coefficients inside the 2^-50 band whose rounding was PROVEN by the exact re-derivation in k_fft_encode (and taken from it) since the last reset
extern "C" int sfg_ctx_encoder_resolved(sfg_ctx *ctx, unsigned long long *count) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    SFG_TRY(sfg_sync_all(ctx));
    SFG_HIP(ctx, hipMemcpy(count, (unsigned long long *)ctx->tie_count_dev + 2, 8, hipMemcpyDeviceToHost));
    return 0;
}
// the sticky 2^-50 counter behind sfg_encoder_check (not reset)
extern "C" int sfg_ctx_encoder_unprovable(sfg_ctx *ctx, unsigned long long *count) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    SFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SFG_HIP(ctx, hipMemcpy(count, (unsigned long long *)ctx->tie_count_dev + 1, 8, hipMemcpyDeviceToHost));
    return 0;
}
extern "C" int sfg_ctx_encoder_inject_unsafe_for_test(sfg_ctx *ctx, unsigned long long n) {      // test hook: pretend n such coefficients were seen
    if (!ctx->test_hooks) SFG_FAIL(ctx, "sfg_ctx_encoder_inject_unsafe_for_test: test hook, enabled only in a process that set the test switch before creating the context");
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    unsigned long long c[2] = {0, n};
    SFG_HIP(ctx, hipMemcpy((unsigned long long *)ctx->tie_count_dev + 1, c + 1, 8, hipMemcpyHostToDevice));
    return 0;
}
