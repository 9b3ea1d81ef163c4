// ntt.hip — negacyclic NTT / inverse NTT over one modulus row of N = 16384 words (lattigo ring.NTT /
// ring.InvNTT restated; they sit behind EncodeNTT at matmult.go:723 and behind every rotation,
// crypto/basics.go:201-224).
//
// One 512-thread workgroup transforms one row.  A row index j = a*512 + b*16 + c (a,b < 32, c < 16):
//   phase A: stages with t >= 512  act on `a`  -> thread (b,c) keeps the 32 values of its column in VGPRs
//   phase B: stages t = 256..16    act on `b`  -> thread (a,c)
//   phase C: stages t = 8..1       act on `c`  -> thread (a,b), two groups of 16 values
// The three phases exchange through one 132 KiB LDS image; HBM sees exactly one coalesced read and one
// coalesced write of the row.  Arithmetic: exact integers in fp64, lazy (values stay in (-2^51, 2^51)),
// one canonical reduction at the end — the canonical output is representation-independent, so it is
// bit-identical to lattigo's Montgomery-form butterflies.
#include "common.hpp"
#include "kernels.hpp"
#include "i8_move.hpp"
#include "ntt_core.hpp"

// IN_MODE 0: rows of canonical u64; 1: half-coefficient input (integer-valued doubles) of a real-slot plaintext
// (pc[0..N/2): p_c, with p_{N/2} = 0 and p_{N-c} = -p_c, see encode.hip), rows = [plain][L]
template <int IN_MODE>
__global__ void __launch_bounds__(512) k_ntt_fwd(const void *in_, u64 *out_, ModPattern pat, RowMap rm, const double *tw_all, const double2 *pack_all, const ModConst *modc) {
    extern __shared__ double lds[];
    const int N = SFG_N, tid = threadIdx.x;
    const size_t row = blockIdx.x;
    const int m = pat.m[row % pat.period];
    const size_t grp = row / rm.rpg, gi = row % rm.rpg;
    if (m == -2) return;                                 // row marked "nobody reads it": neither transformed nor copied
    if (m < 0) {                                         // row marked "leave untouched": copied through when the transform runs out of place
        const u64 *src = (const u64 *)in_ + grp * rm.gstride_in + gi * N; u64 *dst = out_ + grp * rm.gstride_out + gi * N;
        if (IN_MODE == 0 && src != dst) for (int a = 0; a < 32; a++) dst[a * 512 + tid] = src[a * 512 + tid];
        return;
    }
    const double *tw = tw_all + (size_t)m * N;
    const double2 *pack = pack_all + (size_t)m * (N / 2);      // late-stage twiddles, see build_pack() in ctx.hip
    const double q = modc[m].q, qinv = modc[m].qinv;
    double v[32];
    // ---- phase A load: j = a*512 + tid
    if (IN_MODE == 0) {
        const u64 *in = (const u64 *)in_ + grp * rm.gstride_in + gi * N;
#pragma unroll
        for (int a = 0; a < 32; a++) v[a] = u64_to_f64(in[a * 512 + tid]);
    } else {
        const double *pc = (const double *)in_ + (row / pat.period) * (size_t)(N / 2);
#pragma unroll
        for (int a = 0; a < 16; a++) v[a] = pc[a * 512 + tid];
#pragma unroll
        for (int a = 16; a < 32; a++) {                 // j = N/2 + x, x = (a-16)*512 + tid: p_j = -p_{N/2 - x}, p_{N/2} = 0
            int x = (a - 16) * 512 + tid;
            v[a] = x == 0 ? 0.0 : -pc[N / 2 - x];
        }
    }
    ntt_fwd_phases(v, lds, tw, pack, q, qinv, tid);
    u64 *out = out_ + grp * rm.gstride_out + gi * N;
    {
        const int b = tid >> 4, c = tid & 15;
#pragma unroll
        for (int a = 0; a < 32; a++) out[a * 512 + tid] = f64_to_u64(canon(lds[a * LDS_ROW + c * 33 + b], q, qinv));
    }
}

__global__ void __launch_bounds__(512) k_ntt_inv(const u64 *in_, u64 *out_, ModPattern pat, RowMap rm, const double *tw_all, const double2 *pack_all, const ModConst *modc) {
    extern __shared__ double lds[];
    const int N = SFG_N, tid = threadIdx.x;
    const size_t row = blockIdx.x;
    const int m = pat.m[row % pat.period];
    if (m < 0) return;
    const size_t grp = row / rm.rpg, gi = row % rm.rpg;
    const double *tw = tw_all + (size_t)m * N;
    const double2 *pack = pack_all + (size_t)m * (N / 2);      // late-stage twiddles, see build_pack() in ctx.hip
    const double q = modc[m].q, qinv = modc[m].qinv;
    const u64 *in = in_ + grp * rm.gstride_in + gi * N;
    double v[32];
    {   // coalesced load into the (a, c*33 + b) image
        const int b = tid >> 4, c = tid & 15;
#pragma unroll
        for (int a = 0; a < 32; a++) lds[a * LDS_ROW + c * 33 + b] = u64_to_f64(in[a * 512 + tid]);
    }
    __syncthreads();
    // ---- phase C': stages t = 1,2,4,8 on c
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int p = tid + 512 * h, a = p >> 5, b = p & 31;
        double w[16];
#pragma unroll
        for (int c = 0; c < 16; c++) w[c] = lds[a * LDS_ROW + c * 33 + b];
        double tl[16];
        {
            const double2 *pk = pack + (size_t)(p >> 6) * 512 + (p & 63);
#pragma unroll
            for (int i = 0; i < 8; i++) { const double2 e = pk[i * 64]; tl[2 * i] = e.x; tl[2 * i + 1] = e.y; }
        }
        gs_stage<16, 1>(w, q, qinv, [&](int g) { return tl[7 + g]; });
        gs_stage<16, 2>(w, q, qinv, [&](int g) { return tl[3 + g]; });
        gs_stage<16, 4>(w, q, qinv, [&](int g) { return tl[1 + g]; });
        gs_stage<16, 8>(w, q, qinv, [&](int g) { return tl[0 + g]; });
#pragma unroll
        for (int c = 0; c < 16; c++) v[h * 16 + c] = pred(w[c], q, qinv);
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int p = tid + 512 * h, a = p >> 5, b = p & 31;
#pragma unroll
        for (int c = 0; c < 16; c++) lds[a * LDS_ROW + c * 33 + b] = v[h * 16 + c];
    }
    __syncthreads();
    // ---- phase B': stages t = 16..256 on b
    {
        const int a = tid >> 4, c = tid & 15;
#pragma unroll
        for (int b = 0; b < 32; b++) v[b] = lds[a * LDS_ROW + c * 33 + b];
        gs_stage<32, 1>(v, q, qinv, [&](int g) { return tw[512 + a * 16 + g]; });
        gs_stage<32, 2>(v, q, qinv, [&](int g) { return tw[256 + a * 8 + g]; });
        gs_stage<32, 4>(v, q, qinv, [&](int g) { return tw[128 + a * 4 + g]; });
        gs_stage<32, 8>(v, q, qinv, [&](int g) { return tw[64 + a * 2 + g]; });
        gs_stage<32, 16>(v, q, qinv, [&](int g) { return tw[32 + a + g]; });
        __syncthreads();
#pragma unroll
        for (int b = 0; b < 32; b++) lds[a * LDS_ROW + b * 16 + c] = pred(v[b], q, qinv);
    }
    __syncthreads();
    // ---- phase A': stages t = 512..8192 on a
#pragma unroll
    for (int a = 0; a < 32; a++) v[a] = lds[a * LDS_ROW + tid];
    gs_stage<32, 1>(v, q, qinv, [&](int g) { return tw[16 + g]; });
    gs_stage<32, 2>(v, q, qinv, [&](int g) { return tw[8 + g]; });
    gs_stage<32, 4>(v, q, qinv, [&](int g) { return tw[4 + g]; });
    gs_stage<32, 8>(v, q, qinv, [&](int g) { return tw[2 + g]; });
    gs_stage<32, 16>(v, q, qinv, [&](int g) { return tw[1 + g]; });
    const double ninv = modc[m].ninv, ninv_q = modc[m].ninv_q;
    u64 *out = out_ + grp * rm.gstride_out + gi * N;
#pragma unroll
    for (int a = 0; a < 32; a++) {
        double x = mulmod_lazy(pred(v[a], q, qinv), ninv, ninv_q, q);
        out[a * 512 + tid] = f64_to_u64(canon(x, q, qinv));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Half-size forward NTT for plaintexts of REAL slot vectors.  Such a polynomial is invariant under X -> X^-1
// (p_{N-c} = -p_c), hence P[N-1-i] = P[i] in lattigo's output order, and the whole "minus" branch of the first
// Cooley-Tukey stage is redundant.  This kernel computes r_j = p_j + W p_{n+j} = p_j - W p_{n-j} (W = psi^(N/2),
// r_0 = p_0) and runs stages 2..14 on those n = N/2 values only: half the butterflies, half the LDS (two
// workgroups per CU), and it emits half rows P[0..n).  256 threads, j = a*512 + b*16 + c with a < 16.
// Block -> (plaintext, modulus) for the plaintext NTT kernels.  The L workgroups of one plaintext read the same 64 KiB of coefficients; workgroups are
// dealt round-robin over the 8 XCDs, each with its own L2, so they are numbered b, b + 8, ..., b + 8 (L - 1): same XCD, dispatched back to back -
// the coefficients come from HBM once instead of once per modulus (measured with row = blockIdx.x: 4.6x the unique bytes).
__device__ __forceinline__ bool plain_block_of(size_t b, size_t nplain, int L, size_t &row, int &m) {
    const size_t per = (size_t)8 * L, plain = (b / per) * 8 + (b % per) % 8;
    m = (int)((b % per) / 8); row = plain * L + m;
    return plain < nplain;
}
__device__ __forceinline__ bool plain_block(size_t nplain, int L, size_t &row, int &m) { return plain_block_of(blockIdx.x, nplain, L, row, m); }
#ifdef SFG_AB          // the full-image form of the plaintext NTT (round 1): A/B build only
constexpr int HLDS_DOUBLES = 16 * LDS_ROW;   // 67,584 B
__global__ void __launch_bounds__(256) k_ntt_half(const double *pc_all, u64 *out_, size_t nplain, int L, PanelMap pm, const double *tw_all, const double2 *pack_all, const ModConst *modc) {
    extern __shared__ double lds[];
    const int N = SFG_N, n = N / 2, tid = threadIdx.x;
    size_t row; int m;
    if (!plain_block(nplain, L, row, m)) return;
    const double *tw = tw_all + (size_t)m * N;
    const double2 *pack = pack_all + (size_t)m * (N / 2);
    const double q = modc[m].q, qinv = modc[m].qinv;
    const double *pc = pc_all + (row / L) * (size_t)n;
    const double W = tw[1], Wq = W * qinv;
    double v[32];
    // ---- phase A: two (b,c) columns per thread, 16 values of a each; stages t = 4096, 2048, 1024, 512
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int pp = tid + 256 * h;
        double w[16];
#pragma unroll
        for (int a = 0; a < 16; a++) {
            const int j = a * 512 + pp;
            const double lo = pc[j];
            const double hi = j == 0 ? 0.0 : pc[n - j];          // p_{n+j} = -p_{n-j}
            w[a] = lo - mulmod_lazy(hi, W, Wq, q);
        }
        ct_stage<16, 8>(w, q, qinv, [&](int g) { return tw[2 + g]; });
        ct_stage<16, 4>(w, q, qinv, [&](int g) { return tw[4 + g]; });
        ct_stage<16, 2>(w, q, qinv, [&](int g) { return tw[8 + g]; });
        ct_stage<16, 1>(w, q, qinv, [&](int g) { return tw[16 + g]; });
#pragma unroll
        for (int a = 0; a < 16; a++) lds[a * LDS_ROW + pp] = w[a];
    }
    __syncthreads();
    // ---- phase B: thread (a, c), 32 values of b; stages t = 256 .. 16
    {
        const int a = tid >> 4, c = tid & 15;
#pragma unroll
        for (int b = 0; b < 32; b++) v[b] = lds[a * LDS_ROW + b * 16 + c];
        ct_stage<32, 16>(v, q, qinv, [&](int g) { return tw[32 + a + g]; });
        ct_stage<32, 8>(v, q, qinv, [&](int g) { return tw[64 + a * 2 + g]; });
        ct_stage<32, 4>(v, q, qinv, [&](int g) { return tw[128 + a * 4 + g]; });
        ct_stage<32, 2>(v, q, qinv, [&](int g) { return tw[256 + a * 8 + g]; });
        ct_stage<32, 1>(v, q, qinv, [&](int g) { return tw[512 + a * 16 + g]; });
        __syncthreads();
#pragma unroll
        for (int b = 0; b < 32; b++) lds[a * LDS_ROW + c * 33 + b] = v[b];
    }
    __syncthreads();
    // ---- phase C: two (a, b) groups per thread, 16 values of c; stages t = 8 .. 1 with the packed twiddles
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int p = tid + 256 * h, a = p >> 5, b = p & 31;
        double w[16];
#pragma unroll
        for (int c = 0; c < 16; c++) w[c] = lds[a * LDS_ROW + c * 33 + b];
        double tl[16];
        {
            const double2 *pk = pack + (size_t)(p >> 6) * 512 + (p & 63);
#pragma unroll
            for (int i = 0; i < 8; i++) { const double2 e = pk[i * 64]; tl[2 * i] = e.x; tl[2 * i + 1] = e.y; }
        }
        ct_stage<16, 8>(w, q, qinv, [&](int g) { return tl[0 + g]; });
        ct_stage<16, 4>(w, q, qinv, [&](int g) { return tl[1 + g]; });
        ct_stage<16, 2>(w, q, qinv, [&](int g) { return tl[3 + g]; });
        ct_stage<16, 1>(w, q, qinv, [&](int g) { return tl[7 + g]; });
#pragma unroll
        for (int c = 0; c < 16; c++) v[h * 16 + c] = w[c];
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int p = tid + 256 * h, a = p >> 5, b = p & 31;
#pragma unroll
        for (int c = 0; c < 16; c++) lds[a * LDS_ROW + c * 33 + b] = v[h * 16 + c];
    }
    __syncthreads();
    // destination plaintext slot inside a (possibly multi-block-row) panel: see PanelMap
    const size_t plain = row / L; const int shift = pm.shift0 + (int)plain;
    const size_t dst = pm.G ? ((size_t)(shift / SFG_D) * pm.G + pm.g) * SFG_D + (size_t)(shift % SFG_D) : plain;
    u64 *out = out_ + (dst * L + m) * (size_t)n;
    const bool packed = (pm.packed_mask >> m) & 1u;
#pragma unroll
    for (int k = 0; k < 32; k++) {
        const int j = k * 256 + tid, a = j >> 9, x = j & 511, b = x >> 4, c = x & 15;
        const u64 w = f64_to_u64(canon(lds[a * LDS_ROW + c * 33 + b], q, qinv));
        out[j] = packed ? pack_limbs(w) : w;
    }
}
#endif
// Same transform with every exchange split in two rounds through a HALF image (33 KiB instead of 66 KiB): three
// workgroups (12 waves) fit a CU instead of two.  Round structure:
//   A->B  by column half h (pp = tid + 256 h  <=>  b < 16 or b >= 16): everyone writes its column h, (a, c) readers take 16 b's
//   B->C / C->out: wave-private 16 x 16 transposes, one group b = (tid & 15) + 16 r at a time (see below)
constexpr int H3_ROWA = 264;                  // 256 + 8 doubles per `a` row of the A->B half image
constexpr int H3_WREG = 4 * 272;              // doubles per wave-private region (B->C transposes, output staging)
constexpr int H3_DOUBLES = 16 * H3_ROWA > 4 * H3_WREG ? 16 * H3_ROWA : 4 * H3_WREG;
constexpr int H3_LDS_BYTES = H3_DOUBLES * 8;      // 34,816 B: four workgroups per CU
// The 13 stages that follow the first Cooley-Tukey stage, on ONE half (hs = 0: indices [0, n), hs = 1: [n, N)) of a row: after stage 1 the
// halves are independent size-n transforms whose twiddles sit hs * (m / 2) further in each stage's table (tw[m + i], i in [hs m/2, (hs+1) m/2)).
// first(j) returns the stage-1 output r_j of this half, j < n; store(j, x) receives the lazy result of output index hs * n + j.
struct NoFill {};
// WIDE: store(j0, v0, v1, v2, v3) receives four CONSECUTIVE outputs (a lane takes words 4 lane .. 4 lane + 3 of each 256-word run instead of every 64th)
template <class First, class Store, class Fill = NoFill, bool WIDE = false>
__device__ __forceinline__ void ntt_half3_body(int hs, First first, Store store, double *lds, const double *tw, const double2 *pack, double q, double qinv, int tid, Fill fill = Fill()) {
    double v[32];
    const int a_b = tid >> 4, c_b = tid & 15;                  // phase B identity
    // ---- phase A (two columns) interleaved with the two A->B rounds
    // (both columns' stage-1 values are formed up front: all 64 input loads are in flight together and their latency is paid once)
    double w2[2][16];
    if constexpr (std::is_same<Fill, NoFill>::value) {
#pragma unroll
        for (int h = 0; h < 2; h++)
#pragma unroll
            for (int a = 0; a < 16; a++) w2[h][a] = first(a * 512 + tid + 256 * h);
    } else fill(w2);                                           // the stage-1 values arrive some other way (k_ntt_half3<true>); returns after a workgroup barrier
#pragma unroll
    for (int h = 0; h < 2; h++) {
        double (&w)[16] = w2[h];
        ct_stage<16, 8>(w, q, qinv, [&](int g) { return tw[2 + hs + g]; });
        ct_stage<16, 4>(w, q, qinv, [&](int g) { return tw[4 + 2 * hs + g]; });
        ct_stage<16, 2>(w, q, qinv, [&](int g) { return tw[8 + 4 * hs + g]; });
        ct_stage<16, 1>(w, q, qinv, [&](int g) { return tw[16 + 8 * hs + g]; });
        if (h) __syncthreads();                                           // round-0 readers are done with the image
#pragma unroll
        for (int a = 0; a < 16; a++) lds[a * H3_ROWA + tid] = w[a];
        __syncthreads();
#pragma unroll
        for (int b = 0; b < 16; b++) v[h * 16 + b] = lds[a_b * H3_ROWA + b * 16 + c_b];
    }
    // ---- phase B: thread (a, c), 32 values of b; stages t = 256 .. 16
    const int ab = 16 * hs + a_b;
#if defined(SFG_NTT_DIAG) && SFG_NTT_DIAG == 3          // timing only: no phase-B twiddle loads
    ct_stage<32, 16>(v, q, qinv, [&](int g) { return (double)(ab + g + 3); });
    ct_stage<32, 8>(v, q, qinv, [&](int g) { return (double)(ab * 2 + g + 5); });
    ct_stage<32, 4>(v, q, qinv, [&](int g) { return (double)(ab * 4 + g + 7); });
    ct_stage<32, 2>(v, q, qinv, [&](int g) { return (double)(ab * 8 + g + 9); });
    ct_stage<32, 1>(v, q, qinv, [&](int g) { return (double)(ab * 16 + g + 11); });
#else
    ct_stage<32, 16>(v, q, qinv, [&](int g) { return tw[32 + ab + g]; });
    ct_stage<32, 8>(v, q, qinv, [&](int g) { return tw[64 + ab * 2 + g]; });
    ct_stage<32, 4>(v, q, qinv, [&](int g) { return tw[128 + ab * 4 + g]; });
    ct_stage<32, 2>(v, q, qinv, [&](int g) { return tw[256 + ab * 8 + g]; });
    ct_stage<32, 1>(v, q, qinv, [&](int g) { return tw[512 + ab * 16 + g]; });
#endif
    // ---- B->C, phase C and the output staging are WAVE-PRIVATE.  Thread (a_b, c_b) of phase B and the phase-C owner of group (a, b) with
    // a = tid >> 4 live in the same 16-lane quarter of a wave: the B->C exchange is a 16 x 16 transpose inside each quarter, and staging a wave's
    // results for coalesced stores needs only that wave's four `a` rows.  So each wave works in its own 8 KiB of the (dead) A->B image with
    // wave-level ordering only - no workgroup barrier after phase B - and takes its two groups b = (tid & 15) + 16 r one after the other, which
    // keeps 16 + 16 values live instead of 32 + 32 (128 registers: four workgroups per CU).
    __syncthreads();                                                      // the A->B image is dead
    {
        const int wv = tid >> 6, lane = tid & 63, al = lane >> 4, bk = lane & 15;
        double *img = lds + wv * H3_WREG;                                 // [a_local 4][c 16][bk 16 (+1 pad)]: rows of 17 doubles, `a` blocks of 272 (= 16 mod 32):
        auto sw = [](int idx) { return (idx >> 8) * 272 + ((idx >> 4) & 15) * 17 + (idx & 15); };   // conflict-free, and every access is base + immediate
#pragma unroll
        for (int r = 0; r < 2; r++) {
            double wc[16];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");           // (the previous round's staging reads are issued)
#pragma unroll
            for (int k = 0; k < 16; k++) img[sw(al * 256 + c_b * 16 + k)] = v[16 * r + k];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
            for (int c = 0; c < 16; c++) wc[c] = img[sw(al * 256 + c * 16 + bk)];
            const int p = a_b * 32 + 16 * r + bk;                         // group (a, b)
            double tl[16];
            {
                const double2 *pk = pack + (size_t)((p >> 6) + 8 * hs) * 512 + (p & 63);
#pragma unroll
                for (int i = 0; i < 8; i++) {
#if defined(SFG_NTT_DIAG) && SFG_NTT_DIAG == 4          // timing only: no phase-C twiddle loads
                    const double2 e = make_double2((double)(p + 2 * i + 3), (double)(p + 2 * i + 4)); (void)pk;
#else
                    const double2 e = pk[i * 64];
#endif
                    tl[2 * i] = e.x; tl[2 * i + 1] = e.y;
                }
            }
            ct_stage<16, 8>(wc, q, qinv, [&](int g) { return tl[0 + g]; });
            ct_stage<16, 4>(wc, q, qinv, [&](int g) { return tl[1 + g]; });
            ct_stage<16, 2>(wc, q, qinv, [&](int g) { return tl[3 + g]; });
            ct_stage<16, 1>(wc, q, qinv, [&](int g) { return tl[7 + g]; });
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
            for (int c = 0; c < 16; c++) img[sw(al * 256 + c * 16 + bk)] = wc[c];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            // output j = a*512 + (16 r + b')*16 + c: per `a` row 256 consecutive words, 64 lanes at a time
#pragma unroll
            for (int ai = 0; ai < 4; ai++) {
                if constexpr (WIDE) {
                    const int x0 = 4 * lane;                             // x0 + e: b' = lane >> 2, c = 4 (lane & 3) + e
                    const int base = ai * 256 + (lane >> 2);
                    store((4 * wv + ai) * 512 + 256 * r + x0, img[sw(base + (4 * (lane & 3) + 0) * 16)], img[sw(base + (4 * (lane & 3) + 1) * 16)],
                          img[sw(base + (4 * (lane & 3) + 2) * 16)], img[sw(base + (4 * (lane & 3) + 3) * 16)]);
                } else {
#pragma unroll
                    for (int qd = 0; qd < 4; qd++) {
                        const int x = qd * 64 + lane;                        // b' = x >> 4, c = x & 15
                        store((4 * wv + ai) * 512 + 256 * r + x, img[sw(ai * 256 + (x & 15) * 16 + (x >> 4))]);
                    }
                }
            }
        }
    }
}
// PERM: the coefficient rows come from the plaintext cache of the block in its OTHER orientation (matmul.hip): plaintext t of this product is the automorphism
// image X -> X^g of cached row u (perm[t] = u | g << 16): coefficient i of the row becomes coefficient i g mod 2N of the image (>= N: minus coefficient - N), and a
// row is stored as its first N/2 coefficients (p_{N-c} = -p_c, p_{N/2} = 0).  The rounding of the encoder commutes with this signed permutation, so the NTT input is
// the very integer polynomial a fresh encode of the rotated diagonal would give.
// The digit planes are written once and read much later (by the transposition pass, after the whole panel): streaming stores keep them out of the way of the coefficient
// rows the encode FFT has just left in the cache for this kernel - with them a launch pair takes 2048 plaintexts instead of 1024 (profiles/r05_ntt_streaming_stores.txt)
#define NT_ST(p, v) __builtin_nontemporal_store((unsigned)(v), (p))
// (the workgroup's body, for block number vb of a launch over nplain plaintexts: k_ntt_half3, and k_ntt_half3_move where mover workgroups come first in the grid)
template <bool PERM, bool DIG>
__device__ __forceinline__ void ntt_half3_wg(size_t vb, double *lds, const double *pc_all, u64 *out_, size_t nplain, int L, PanelMap pm, const double *tw_all, const double2 *pack_all,
                                             const ModConst *modc, const uint32_t *perm) {
    const int N = SFG_N, n = N / 2, tid = threadIdx.x;
    size_t row; int m;
    if (!plain_block_of(vb, nplain, L, row, m)) return;
    const double *tw = tw_all + (size_t)m * N;
    const double2 *pack = pack_all + (size_t)m * (N / 2);
    const double q = modc[m].q, qinv = modc[m].qinv;
    const double *pc = pc_all + (row / L) * (size_t)n;
    unsigned gal = 1;
    if (PERM) { const uint32_t e = perm[pm.shift0 + (int)(row / L)]; pc = pc_all + (size_t)(e & 0xFFFFu) * n; gal = e >> 16; }
    const double W = tw[1], Wq = W * qinv;
    // stage 1 on the antisymmetric input: r_j = p_j + W p_{n+j} = p_j - W p_{n-j}, r_0 = p_0
#if defined(SFG_NTT_DIAG) && SFG_NTT_DIAG == 5              // timing only: no coefficient-row loads
    auto first = [&](int j) { const double lo = (double)(j + 7), hi = j == 0 ? 0.0 : (double)(n - j + 5); return lo - mulmod_lazy(hi, W, Wq, q); };
#else
    auto first = [&](int j) { const double lo = pc[j], hi = j == 0 ? 0.0 : pc[n - j]; return lo - mulmod_lazy(hi, W, Wq, q); };
#endif
    // PERM: source driven.  A thread reads the pairs (p_i, p_{n-i}), i = tid + 256 k < n/2, coalesced.  p_i is coefficient raw = i g mod 2N of the image, i.e.
    // +-coefficient J of its stored half (quadrants of raw: [0,n] J = raw, +; (n,N) J = N - raw, -; [N,N+n] J = raw - N, -; (N+n,2N) J = 2N - raw, +), and
    // because g = 1 mod 4 its partner p_{n-i} is coefficient n - J (signs +, +, -, -): the pair yields r_J and r_{n-J}.  The 8192 stage-1 values then go to
    // their phase-A owners through the (not yet used) A->B image, one column half per round: odd g makes the scattered writes bank-conflict free.
    auto fill = [&](double (&w2)[2][16]) {
        double rA[16], rB[16];
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const int i = tid + 256 * k;
            const double lo = pc[i], hi = i == 0 ? 0.0 : pc[n - i];
            const unsigned qd = (((unsigned)i * gal) & (2u * N - 1u)) >> 13;
            const double ca = (qd == 1u || qd == 2u) ? -lo : lo, cb = qd >= 2u ? -hi : hi;
            rA[k] = ca - mulmod_lazy(cb, W, Wq, q);
            rB[k] = cb - mulmod_lazy(ca, W, Wq, q);
        }
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int k = 0; k < 16; k++) {
                const unsigned raw = ((unsigned)(tid + 256 * k) * gal) & (2u * N - 1u), qd = raw >> 13;
                const int J = qd == 0u ? (int)raw : qd == 1u ? N - (int)raw : qd == 2u ? (int)raw - N : 2 * N - (int)raw, Jb = n - J;
                if (((J >> 8) & 1) == h && J < n) lds[(J >> 9) * H3_ROWA + (J & 255)] = rA[k];
                if (((Jb >> 8) & 1) == h && Jb < n) lds[(Jb >> 9) * H3_ROWA + (Jb & 255)] = rB[k];
            }
            __syncthreads();
#pragma unroll
            for (int a = 0; a < 16; a++) w2[h][a] = lds[a * H3_ROWA + tid];
            __syncthreads();
        }
        if (tid == 0) {                                          // the self-paired middle coefficient p_{n/2} -> r_{n/2}: thread 0's own slot (a = 8, column 0)
            const double mid = pc[n / 2];
            const double c = ((((unsigned)(n / 2) * gal) & (2u * N - 1u)) >> 13) >= 2u ? -mid : mid;
            w2[0][8] = c - mulmod_lazy(c, W, Wq, q);
        }
    };
    // destination plaintext slot inside a (possibly multi-block-row) panel: see PanelMap
    const size_t plain = row / L; const int shift = pm.shift0 + (int)plain;
    const size_t dst = pm.G ? ((size_t)(shift / SFG_D) * pm.G + pm.g) * SFG_D + (size_t)(shift % SFG_D) : plain;
    u64 *out = out_ + (dst * L + m) * (size_t)n;
    size_t dstride = n, cbm = 0;       // bytes between the digit planes of this row; K-major: (bytes between 128-byte coefficient blocks) - 128, so that byte j of a plane is at j + (j >> 7) * cbm
    if constexpr (DIG) if (pm.packed_mask & PT_COMPACT) {      // compact panel rows: the moduli's 5 / 6 digit planes of a plaintext back to back (bit l of the mask: modulus l has five)
        const unsigned lm = (1u << L) - 1u, small = pm.packed_mask & lm;
        const int ns_all = __popc(small), ns_below = __popc(small & ((1u << m) - 1u));
        const size_t planes_all = (size_t)ns_all * 5 + (size_t)(L - ns_all) * 6, planes_below = (size_t)ns_below * 5 + (size_t)(m - ns_below) * 6;
        if (pm.packed_mask & PT_KMAJOR) {
            const size_t K = (size_t)pm.K, col = pm.G ? (size_t)(shift / SFG_D) : plain / K, row = pm.G ? (size_t)pm.g * SFG_D + (size_t)(shift % SFG_D) : plain % K;
            out = reinterpret_cast<u64 *>(reinterpret_cast<uint8_t *>(out_) + ((col * planes_all + planes_below) * 64 * K + row) * 128);
            dstride = 64 * K * 128; cbm = K * 128 - 128;
        } else out = reinterpret_cast<u64 *>(reinterpret_cast<uint8_t *>(out_) + dst * (planes_all * n) + planes_below * n);
    }
    // (the format test is hoisted: inside the store loop it costs a branch per word)
#ifdef SFG_NTT_DIAG          // timing diagnostics only: 1 = no panel stores (kept alive by an impossible value), 2 = stores without canon / packing
    if (SFG_NTT_DIAG == 1) { ntt_half3_body(0, first, [&](int j, double x) { if (x == 0.123) out[j] = pack_limbs_f64(canon_le(x, q, qinv)); }, lds, tw, pack, q, qinv, tid); return; }
    if (SFG_NTT_DIAG == 2) { ntt_half3_body(0, first, [&](int j, double x) { out[j] = (u64)__double_as_longlong(x); }, lds, tw, pack, q, qinv, tid); return; }
#endif
    // int8 MAC (mac_i8.hip; bit 31 of the mask): the packed rows leave as five planes of signed base-256 digits instead of words - 5 of the row's 8 bytes per word.
    // Balanced digits of v: the bytes of v + 0x8080808080 with their top bits flipped; the sum is read off the mantissa of v + 2^52 + 0x8080808080.
    if constexpr (DIG) if (!((pm.packed_mask >> m) & 1u) && ((pm.packed_mask >> 30) & 1u)) {        // the 46-bit row as SIX digit planes (48 KiB of its 64 KiB)
        uint8_t *o8 = reinterpret_cast<uint8_t *>(out);
        auto dig6 = [&](double x, unsigned &lo, unsigned &hi) {
            const u64 b = (u64)__double_as_longlong(canon(x, q, qinv) + (4503599627370496.0 + 141289400074368.0));       // + 2^52 + 0x808080808080
            lo = (unsigned)b ^ 0x80808080u; hi = (unsigned)(b >> 32) ^ 0x8080u;
        };
        auto st6 = [&](int j0, double x0, double x1, double x2, double x3) {
            unsigned l0, l1, l2, l3, h0, h1, h2, h3;
            dig6(x0, l0, h0); dig6(x1, l1, h1); dig6(x2, l2, h2); dig6(x3, l3, h3);
            unsigned o[4]; bytes_tr4(l0, l1, l2, l3, o);
            uint8_t *ob = o8 + j0 + (size_t)(j0 >> 7) * cbm;
#pragma unroll
            for (int d = 0; d < 4; d++) NT_ST(reinterpret_cast<unsigned *>(ob + d * dstride), o[d]);
            unsigned p[4]; bytes_tr4(h0, h1, h2, h3, p);
            NT_ST(reinterpret_cast<unsigned *>(ob + 4 * dstride), p[0]);
            NT_ST(reinterpret_cast<unsigned *>(ob + 5 * dstride), p[1]);
        };
        if constexpr (PERM) ntt_half3_body<decltype(first), decltype(st6), decltype(fill), true>(0, first, st6, lds, tw, pack, q, qinv, tid, fill);
        else ntt_half3_body<decltype(first), decltype(st6), NoFill, true>(0, first, st6, lds, tw, pack, q, qinv, tid);
        return;
    }
    if constexpr (DIG) if ((pm.packed_mask >> m) & 1u) {
        uint8_t *o8 = reinterpret_cast<uint8_t *>(out);
        auto dig = [&](double x, unsigned &lo, unsigned &hi) {
            const u64 b = (u64)__double_as_longlong(canon_le(x, q, qinv) + (4503599627370496.0 + 551911719040.0));
            lo = (unsigned)b ^ 0x80808080u; hi = (unsigned)(b >> 32) ^ 0x80u;
        };
        auto st8 = [&](int j0, double x0, double x1, double x2, double x3) {        // four consecutive coefficients: one dword per digit plane
            unsigned l0, l1, l2, l3, h0, h1, h2, h3;
            dig(x0, l0, h0); dig(x1, l1, h1); dig(x2, l2, h2); dig(x3, l3, h3);
            unsigned o[4]; bytes_tr4(l0, l1, l2, l3, o);
            uint8_t *ob = o8 + j0 + (size_t)(j0 >> 7) * cbm;
#pragma unroll
            for (int d = 0; d < 4; d++) NT_ST(reinterpret_cast<unsigned *>(ob + d * dstride), o[d]);
            NT_ST(reinterpret_cast<unsigned *>(ob + 4 * dstride), (h0 & 255u) | ((h1 & 255u) << 8) | ((h2 & 255u) << 16) | (h3 << 24));
        };
        if constexpr (PERM) ntt_half3_body<decltype(first), decltype(st8), decltype(fill), true>(0, first, st8, lds, tw, pack, q, qinv, tid, fill);
        else ntt_half3_body<decltype(first), decltype(st8), NoFill, true>(0, first, st8, lds, tw, pack, q, qinv, tid);
        return;
    }
    if constexpr (PERM) {
        if (!DIG && ((pm.packed_mask >> m) & 1u)) ntt_half3_body(0, first, [&](int j, double x) { out[j] = pack_limbs_f64(canon_le(x, q, qinv)); }, lds, tw, pack, q, qinv, tid, fill);
        else ntt_half3_body(0, first, [&](int j, double x) { out[j] = f64_to_u64(canon(x, q, qinv)); }, lds, tw, pack, q, qinv, tid, fill);
    } else {
        (void)fill;
        if (!DIG && ((pm.packed_mask >> m) & 1u)) ntt_half3_body(0, first, [&](int j, double x) { out[j] = pack_limbs_f64(canon_le(x, q, qinv)); }, lds, tw, pack, q, qinv, tid);
        else ntt_half3_body(0, first, [&](int j, double x) { out[j] = f64_to_u64(canon(x, q, qinv)); }, lds, tw, pack, q, qinv, tid);
    }
}
template <bool PERM, bool DIG>
__global__ void __launch_bounds__(256, 4) k_ntt_half3(const double *pc_all, u64 *out_, size_t nplain, int L, PanelMap pm, const double *tw_all, const double2 *pack_all, const ModConst *modc,
                                                      const uint32_t *perm) {
    extern __shared__ double lds[];
    ntt_half3_wg<PERM, DIG>(blockIdx.x, lds, pc_all, out_, nplain, L, pm, tw_all, pack_all, modc, perm);
}
// The same launch with MOVER workgroups in front (i8_move.hpp): the first job.nblocks workgroups of the grid - dispatched first, one to a CU while the CUs are empty -
// transpose a slice of the PREVIOUS MAC launch's plaintext panel into the int8 MAC's tiles while the NTT workgroups behind them fill the other three slots of every CU.
// The NTT is fp64-issue bound and leaves two thirds of the HBM rate idle; the mover is HBM bound and needs 8 v_perm per 16 bytes.  A launch has ONE LDS size and ONE
// register budget, so a mover workgroup lives in the NTT's 34 KiB and 128 VGPRs; the NTT's block -> XCD numbering is kept (nblocks is a multiple of 8).
template <bool PERM, int DEPTH, bool NT>
__global__ void __launch_bounds__(256, 4) k_ntt_half3_move(const double *pc_all, u64 *out_, size_t nplain, int L, PanelMap pm, const double *tw_all, const double2 *pack_all, const ModConst *modc,
                                                           const uint32_t *perm, MoveJob job) {
    extern __shared__ double lds[];
    if (blockIdx.x < job.nblocks) { i8_move_block<DEPTH, NT>(job, blockIdx.x, reinterpret_cast<unsigned *>(lds), (int)threadIdx.x); return; }
    ntt_half3_wg<PERM, true>((size_t)blockIdx.x - job.nblocks, lds, pc_all, out_, nplain, L, pm, tw_all, pack_all, modc, perm);
}
// Forward NTT of general rows as TWO such workgroups per row (the key switch, Rescale, the bootstrap shares): 256 threads and 33 KiB each, three to a
// CU, instead of one 512-thread workgroup holding a 132 KiB image.  Each half reads both halves of the input (the second read is an L2 hit: the two
// workgroups of a row are numbered b and b + 8, same XCD) and pays the stage-1 product itself.
__global__ void __launch_bounds__(256, 4) k_ntt_fwd_split(const u64 *in_, u64 *out_, size_t nrows, ModPattern pat, RowMap rm, const double *tw_all, const double2 *pack_all, const ModConst *modc) {
    extern __shared__ double lds[];
    const int N = SFG_N, n = N / 2, tid = threadIdx.x;
    const size_t b = blockIdx.x, row = (b / 16) * 8 + b % 8; const int hs = (int)((b / 8) & 1);
    if (row >= nrows) return;
    const int m = pat.m[row % pat.period];
    const size_t grp = row / rm.rpg, gi = row % rm.rpg;
    const u64 *in = in_ + grp * rm.gstride_in + gi * N;
    u64 *out = out_ + grp * rm.gstride_out + gi * N + (size_t)hs * n;
    if (m == -2) return;                                 // row marked "nobody reads it"
    if (m < 0) {                                         // row marked "leave untouched": out of place that means "copy through"
        for (int k = 0; k < 32; k++) out[k * 256 + tid] = in[(size_t)hs * n + k * 256 + tid];
        return;
    }
    const double *tw = tw_all + (size_t)m * N;
    const double2 *pack = pack_all + (size_t)m * (N / 2);
    const double q = modc[m].q, qinv = modc[m].qinv;
    const double W = hs ? -tw[1] : tw[1], Wq = W * qinv;
    auto first = [&](int j) { return u64_to_f64(in[j]) + mulmod_lazy(u64_to_f64(in[n + j]), W, Wq, q); };
    ntt_half3_body(hs, first, [&](int j, double x) { out[j] = f64_to_u64(canon(x, q, qinv)); }, lds, tw, pack, q, qinv, tid);
}
// full rows from half rows: out[i] = out[N-1-i] = half[i]
__global__ void __launch_bounds__(256) k_expand_half(const u64 *half, u64 *full) {
    const int N = SFG_N, n = N / 2; const size_t row = blockIdx.x / (n / 256);
    const int i = (int)(blockIdx.x % (n / 256)) * 256 + threadIdx.x;
    const u64 v = half[row * n + i];
    full[row * N + i] = v; full[row * N + (N - 1 - i)] = v;
}

int ntt_set_attrs(sfg_ctx *ctx) {
    hipError_t e = hipFuncSetAttribute((const void *)k_ntt_fwd<0>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DOUBLES * 8);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_ntt_fwd<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DOUBLES * 8);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_ntt_inv, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DOUBLES * 8);
#ifdef SFG_AB
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_ntt_half, hipFuncAttributeMaxDynamicSharedMemorySize, HLDS_DOUBLES * 8);
#endif
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_ntt_half3<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, H3_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_ntt_half3<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, H3_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_ntt_half3<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, H3_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_ntt_half3<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, H3_LDS_BYTES);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_ntt_fwd_split, hipFuncAttributeMaxDynamicSharedMemorySize, H3_LDS_BYTES);
#define SFG_MV_ATTR(P, D, T) if (e == hipSuccess) e = hipFuncSetAttribute((const void *)k_ntt_half3_move<P, D, T>, hipFuncAttributeMaxDynamicSharedMemorySize, H3_LDS_BYTES)
    SFG_MV_ATTR(false, 1, true); SFG_MV_ATTR(true, 1, true);              // the product's riding transposition: one unit in flight, streaming loads and stores
#ifdef SFG_AB
    SFG_MV_ATTR(false, 1, false); SFG_MV_ATTR(false, 2, false); SFG_MV_ATTR(false, 2, true); SFG_MV_ATTR(false, 3, false); SFG_MV_ATTR(false, 3, true);
    SFG_MV_ATTR(true, 1, false); SFG_MV_ATTR(true, 2, false); SFG_MV_ATTR(true, 2, true); SFG_MV_ATTR(true, 3, false); SFG_MV_ATTR(true, 3, true);
#endif
#undef SFG_MV_ATTR
    if (e != hipSuccess) SFG_FAIL(ctx, "cannot raise dynamic LDS limit for the NTT kernels");
    return 0;
}

static RowMap dense_map() { RowMap rm; rm.rpg = 1; rm.gstride_in = SFG_N; rm.gstride_out = SFG_N; return rm; }
int launch_ntt_fwd(sfg_ctx *ctx, const u64 *in, u64 *out, size_t nrows, const ModPattern &pat) { return launch_ntt_fwd_map(ctx, in, out, nrows, pat, dense_map()); }
int launch_ntt_inv(sfg_ctx *ctx, const u64 *in, u64 *out, size_t nrows, const ModPattern &pat) { return launch_ntt_inv_map(ctx, in, out, nrows, pat, dense_map()); }
int launch_ntt_fwd_map(sfg_ctx *ctx, const u64 *in, u64 *out, size_t nrows, const ModPattern &pat, const RowMap &rm) {
    if (!nrows) return 0;
    // in place the two half-workgroups of a row would overwrite each other's input: the split form needs distinct buffers
    if (in != out && !ctx->cfg.ntt_fwd_full)
        hipLaunchKernelGGL(k_ntt_fwd_split, dim3((unsigned)((nrows + 7) / 8 * 16)), dim3(256), H3_LDS_BYTES, ctx->stream, in, out, nrows, pat, rm, ctx->tw_fwd, ctx->pack_fwd, ctx->modc);
    else
        hipLaunchKernelGGL(k_ntt_fwd<0>, dim3((unsigned)nrows), dim3(512), LDS_DOUBLES * 8, ctx->stream, (const void *)in, out, pat, rm, ctx->tw_fwd, ctx->pack_fwd, ctx->modc);
    SFG_HIP(ctx, hipGetLastError());
    return 0;
}
int launch_ntt_plain(sfg_ctx *ctx, const double *pc, u64 *out, size_t nplain, int L) {
    if (!nplain) return 0;
    ModPattern pat; pat.period = L; for (int l = 0; l < L; l++) pat.m[l] = (int8_t)l;
    hipLaunchKernelGGL(k_ntt_fwd<1>, dim3((unsigned)(nplain * L)), dim3(512), LDS_DOUBLES * 8, ctx->stream, (const void *)pc, out, pat, dense_map(), ctx->tw_fwd, ctx->pack_fwd, ctx->modc);
    SFG_HIP(ctx, hipGetLastError());
    return 0;
}
// half rows [nplain][L][N/2] from half-coefficient plaintexts
template <bool PERM>
static void launch_half3_move(sfg_ctx *ctx, dim3 grid, const double *pc, u64 *out_half, size_t nplain, int L, PanelMap pm, const uint32_t *perm, const MoveJob &j) {
#define SFG_MV(D, T) hipLaunchKernelGGL((k_ntt_half3_move<PERM, D, T>), grid, dim3(256), H3_LDS_BYTES, ctx->stream, pc, out_half, nplain, L, pm, ctx->tw_fwd, ctx->pack_fwd, ctx->modc, perm, j)
#ifdef SFG_AB
    if (j.depth == 3) { if (j.nt) SFG_MV(3, true); else SFG_MV(3, false); return; }
    if (j.depth == 2) { if (j.nt) SFG_MV(2, true); else SFG_MV(2, false); return; }
    if (!j.nt) { SFG_MV(1, false); return; }
#endif
    SFG_MV(1, true);
#undef SFG_MV
}
int launch_ntt_plain_half(sfg_ctx *ctx, const double *pc, u64 *out_half, size_t nplain, int L, PanelMap pm, const uint32_t *perm, const MoveJob *mv) {
    if (!nplain) return 0;
    const dim3 grid((unsigned)((nplain + 7) / 8 * 8 * L));
    const bool dig = pm.packed_mask >> 31;                 // digit planes for the int8 MAC (mac_i8.hip): its own instances, the default kernels are untouched
    if (mv && mv->count) {
        if (!dig || mv->nblocks % 8 || !mv->nblocks) SFG_FAIL(ctx, "plaintext NTT: mover workgroups need the digit-plane form and a multiple of 8 of them");
#ifndef SFG_AB
        if (mv->depth != 1 || !mv->nt) SFG_FAIL(ctx, "plaintext NTT: the product build holds the mover with one unit in flight and streaming accesses only (other forms: make ab)");
#endif
        const dim3 g2(grid.x + mv->nblocks);
        if (perm) launch_half3_move<true>(ctx, g2, pc, out_half, nplain, L, pm, perm, *mv);
        else launch_half3_move<false>(ctx, g2, pc, out_half, nplain, L, pm, (const uint32_t *)nullptr, *mv);
        SFG_HIP(ctx, hipGetLastError());
        return 0;
    }
    if (perm && dig) hipLaunchKernelGGL((k_ntt_half3<true, true>), grid, dim3(256), H3_LDS_BYTES, ctx->stream, pc, out_half, nplain, L, pm, ctx->tw_fwd, ctx->pack_fwd, ctx->modc, perm);
    else if (perm) hipLaunchKernelGGL((k_ntt_half3<true, false>), grid, dim3(256), H3_LDS_BYTES, ctx->stream, pc, out_half, nplain, L, pm, ctx->tw_fwd, ctx->pack_fwd, ctx->modc, perm);
    else if (dig) hipLaunchKernelGGL((k_ntt_half3<false, true>), grid, dim3(256), H3_LDS_BYTES, ctx->stream, pc, out_half, nplain, L, pm, ctx->tw_fwd, ctx->pack_fwd, ctx->modc, (const uint32_t *)nullptr);
#ifdef SFG_AB
    else if (ctx->cfg.ntt_half_full) hipLaunchKernelGGL(k_ntt_half, grid, dim3(256), HLDS_DOUBLES * 8, ctx->stream, pc, out_half, nplain, L, pm, ctx->tw_fwd, ctx->pack_fwd, ctx->modc);
#endif
    else hipLaunchKernelGGL((k_ntt_half3<false, false>), grid, dim3(256), H3_LDS_BYTES, ctx->stream, pc, out_half, nplain, L, pm, ctx->tw_fwd, ctx->pack_fwd, ctx->modc, (const uint32_t *)nullptr);
    SFG_HIP(ctx, hipGetLastError());
    return 0;
}
int launch_expand_half(sfg_ctx *ctx, const u64 *half, u64 *full, size_t nrows) {
    if (!nrows) return 0;
    hipLaunchKernelGGL(k_expand_half, dim3((unsigned)(nrows * (SFG_N / 2 / 256))), dim3(256), 0, ctx->stream, half, full);
    SFG_HIP(ctx, hipGetLastError());
    return 0;
}
int launch_ntt_inv_map(sfg_ctx *ctx, const u64 *in, u64 *out, size_t nrows, const ModPattern &pat, const RowMap &rm) {
    if (!nrows) return 0;
    hipLaunchKernelGGL(k_ntt_inv, dim3((unsigned)nrows), dim3(512), LDS_DOUBLES * 8, ctx->stream, in, out, pat, rm, ctx->tw_inv, ctx->pack_inv, ctx->modc);
    SFG_HIP(ctx, hipGetLastError());
    return 0;
}

static int pattern_from_host(sfg_ctx *ctx, const int *mod_idx, int nrows, ModPattern &pat) {
    for (int period = 1; period <= 64 && period <= nrows; period++) {
        bool ok = true;
        for (int r = 0; r < nrows && ok; r++) ok = mod_idx[r] == mod_idx[r % period];
        if (ok) { pat.period = period; for (int i = 0; i < period; i++) { if (mod_idx[i] < 0 || mod_idx[i] >= ctx->nmod) SFG_FAIL(ctx, "modulus index out of range"); pat.m[i] = (int8_t)mod_idx[i]; } return 0; }
    }
    SFG_FAIL(ctx, "mod_idx must be periodic with period <= 64");
}

extern "C" int sfg_ntt_rows(sfg_ctx *ctx, uint64_t *rows, int nrows, const int *mod_idx) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    ModPattern pat; SFG_TRY(pattern_from_host(ctx, mod_idx, nrows, pat));
    return launch_ntt_fwd(ctx, (const u64 *)rows, (u64 *)rows, nrows, pat);
}
extern "C" int sfg_intt_rows(sfg_ctx *ctx, uint64_t *rows, int nrows, const int *mod_idx) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    ModPattern pat; SFG_TRY(pattern_from_host(ctx, mod_idx, nrows, pat));
    return launch_ntt_inv(ctx, (const u64 *)rows, (u64 *)rows, nrows, pat);
}
