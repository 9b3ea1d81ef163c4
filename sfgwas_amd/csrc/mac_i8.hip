// The ring MAC on the int8 matrix core: the default for the 35-bit moduli (SFG_MAC_IMPL=bc restores the fp64 kernel), an option for the 46-bit one (SFG_MAC_I8_BIG=1).
// Same contract as the launches of launch_mac_bc: out[n][r][l][x] (+)= sum_k pt[k][n][l][x'] * rotf[k][r][plane l][x] mod q_l, x' = x or N-1-x.
//
// Per coefficient x that is a 32 x 96 x K GEMM over exact integers.  Both operands are written as ND signed base-256 digits (ND = 5: pt canonical < 2^36, rot centred;
// ND = 6 for the 46-bit modulus), the ND^2 digit products of a k-step go through v_mfma_i32_16x16x64_i8 into 2 ND - 1 int32 sums per output - one per digit-weight
// a + b, each |sum| <= ND K 2^14 < 2^31 - and the sums are recombined mod q by Horner in the epilogue.  Nothing is rounded anywhere.
//
// The matrix instruction contracts over k, so a lane needs 16 consecutive k of ONE coefficient; the product's operands are coefficient-contiguous (a plaintext NTT
// owns all coefficients of one k).  Hence transposition kernels into MFMA register order (1 KiB = 64 lanes x 16 bytes per operand tile, chunk of 64 k and digit):
//   k_i8_pack_rot<ND>        rotf planes         -> A [m][x < N][ch][rt 2][a ND][1 KiB]    once per rot operand generation (a group's rotation cache serves every column)
//   k_i8_pack_pt_digits<ND>  NTT digit planes    -> B [m][c < N/2][jt][ch][b ND][1 KiB]    once per launch (k_i8_pack_pt<ND>: the same from panel words, DiagCache products)
// and the MAC proper (k_mac_i8<ND>) streams both from global memory without LDS: a wave owns one coefficient pair (c, N-1-c share the plaintext word: the pt tile is
// loaded once for 64 rows) x 16 columns = 4 row tiles x (2 ND - 1) weights of accumulator tiles; the <= 6 column waves of a pair form a workgroup and share the rot
// tiles through the cache (k_mac_i8_lds stages them through LDS instead: measured slower).  Results leave in tile order (T [m][c][half][jt][rt][lane][4]) and
// k_i8_untile adds them into the canonical accumulators with coefficient-contiguous runs.
#include "common.hpp"
#include "kernels.hpp"
#include "i8_move.hpp"            // I8Args, I8_PD and the low-occupancy form of the plaintext transposition
#include <algorithm>
#include <cstring>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int I8_ND = 5;                   // digits per operand word of a 35-bit modulus (the 46-bit one: 6)

// ND signed base-256 digits of an integer |v| < 2^(8 ND - 1) (two's complement arithmetic shift)
template <int ND> __device__ __forceinline__ void i8_digits(long long v, int8_t d[ND]) {
#pragma unroll
    for (int i = 0; i < ND; i++) { const long long lo = ((v + 128) & 255) - 128; d[i] = (int8_t)lo; v = (v - lo) >> 8; }
}
// byte offset of element (row-or-column i < 16, kk < 64) inside a 1 KiB operand tile: lane = i + 16 (kk / 16), byte kk % 16
__device__ __forceinline__ int i8_tile_off(int i, int kk) { return ((i + 16 * (kk >> 4)) << 4) + (kk & 15); }

// sum_s D_s 256^s mod q by Horner on exact integers held in fp64: |r| <= q/2 and |D| < 2^31.  One step x = r 256 + D, r' = x - q rint(x / q) is exact while
// 128 q + 2^31 < 2^53 (x itself AND the product q rint(x / q) <= 128 q), i.e. for q <= 2^46 - 2^24: PN14QP438's q0 = 0x200000440001 < 2^46 and every 35-bit prime.
// A modulus in (2^46 - 2^24, 2^47) - accepted by sfg_ctx_create - takes the step as two multiplications by 16 with a reduction in between (8 q + 2^31 < 2^51).
// `wide` is uniform over the launch's modulus (a scalar branch).
constexpr double I8_WIDE_Q = 0x1p46 - 0x1p24;
template <int NS> __device__ __forceinline__ double i8_horner(const v4i (&a)[NS], int e, double q, double qinv, bool wide) {
    double r = (double)a[NS - 1][e];
    if (!wide) {
#pragma unroll
        for (int s = NS - 2; s >= 0; s--) { const double x = r * 256.0 + (double)a[s][e]; r = x - q * __builtin_rint(x * qinv); }
    } else {
#pragma unroll
        for (int s = NS - 2; s >= 0; s--) {
            const double x1 = r * 16.0, r1 = x1 - q * __builtin_rint(x1 * qinv);
            const double x = r1 * 16.0 + (double)a[s][e]; r = x - q * __builtin_rint(x * qinv);
        }
    }
    return r < 0 ? r + q : r;
}

// ---- rot planes -> A.  workgroup = (modulus m, chunk ch, half chunk kh, 16 coefficients, row tile rt): 32 k x 16 rows x 16 coefficients through a 40 (48) KiB digit
// image, four (three) workgroups per CU.  Round 6: 128-byte source runs (16 coefficients; until then 8 coefficients x 64 k x both row tiles through 80 KiB - two
// workgroups per CU whose load and store phases hardly overlapped, 64-byte runs: 2.1 TB/s) with all 32 loads of a thread in flight, and a thread takes FOUR consecutive
// k of its (coefficient, row) so that a digit of the four is one dword of the tile: five dword stores per four elements instead of twenty byte stores, lanes = 16
// coefficients x the 4 dwords of a 16-byte lane row on 64 distinct banks (a coefficient's lane rows are XORed with its number: no padding).
// ND = 6: the 46-bit modulus, whose fp64 plane holds the signed split {lo 23 bits, hi} per coefficient (k_rot_to_f64): v = hi 2^23 + lo
constexpr int I8_PC = 16;                  // coefficients per rot packing workgroup
template <int ND> constexpr int i8_rot_img_stride() { return ND * 512; }              // bytes per coefficient in the image: ND digits x 32 lane rows x 16 bytes
template <int ND>
__global__ void __launch_bounds__(256) k_i8_pack_rot(I8Args a) {
    extern __shared__ __attribute__((aligned(16))) int8_t img[];         // [cc 16][a ND][lane row (32) ^ cc][16 bytes]
    constexpr int CS = i8_rot_img_stride<ND>();
    const int N = SFG_N, tid = threadIdx.x;
    int b = blockIdx.x;
    const int rt = b & 1; b >>= 1;
    const int khalf = b & 1; b >>= 1;
    const int xb = b % (N / I8_PC), ch = (b / (N / I8_PC)) % a.nch, m = b / (N / I8_PC) / a.nch;
    const int x0 = xb * I8_PC, cc = tid & (I8_PC - 1), kkq = (tid >> 4) & 3, kh2 = (tid >> 6) & 1, rh = tid >> 7;
    const double *src = ND == 6 ? a.rotf + (size_t)a.plane0 * N + 2 * (x0 + cc) : a.rotf + (size_t)(a.plane0 + m) * N + x0 + cc;
    // the four k of this thread: ch * 64 + 32 khalf + 16 kh2 + 4 kkq + j (the same for every row)
    size_t koff[4]; bool kv[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int k = ch * 64 + 32 * khalf + 16 * kh2 + 4 * kkq + j;
        int ks = k; kv[j] = k < a.K;
        if (a.kb) { const int gg = k / a.kb, baby = k - gg * a.kb; kv[j] = kv[j] && baby < SFG_D; ks = gg * SFG_D + baby; }
        koff[j] = kv[j] ? (size_t)ks * a.rotf_k_stride : 0;
    }
    double lo[8][4], hi[ND == 6 ? 8 : 1][4];
#pragma unroll
    for (int ri = 0; ri < 8; ri++) {
        const int r = a.r0 + rt * 16 + rh * 8 + ri;
        const double *e = src + (size_t)(r < a.R ? r : a.r0) * a.rotf_r_stride;
#pragma unroll
        for (int j = 0; j < 4; j++) { lo[ri][j] = e[koff[j]]; if (ND == 6) hi[ri][j] = e[koff[j] + 1]; }
    }
#pragma unroll
    for (int ri = 0; ri < 8; ri++) {
        const int r16 = rh * 8 + ri;
        const bool rv = a.r0 + rt * 16 + r16 < a.R;
        unsigned dw[ND];
#pragma unroll
        for (int i = 0; i < ND; i++) dw[i] = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            long long v = ND == 6 ? (long long)hi[ND == 6 ? ri : 0][j] * 8388608LL + (long long)lo[ri][j] : (long long)lo[ri][j];
            if (!(rv && kv[j])) v = 0;
            int8_t d[ND]; i8_digits<ND>(v, d);
#pragma unroll
            for (int i = 0; i < ND; i++) dw[i] |= (unsigned)(uint8_t)d[i] << (8 * j);
        }
        // lane row of (row r16, kk = 16 kh2 + 4 kkq + j) inside this half chunk: r16 + 16 kh2; its 16 bytes hold kk & 15 = 4 kkq + j
        unsigned *o = reinterpret_cast<unsigned *>(img + cc * CS + (((r16 + 16 * kh2) ^ cc) << 4) + 4 * kkq);
#pragma unroll
        for (int i = 0; i < ND; i++) o[i * 128] = dw[i];
    }
    __syncthreads();
    // per coefficient and digit the half chunk is 512 contiguous bytes of the 1 KiB tile (lanes 32 khalf .. + 31)
    for (int idx = tid; idx < I8_PC * ND * 32; idx += 256) {
        const int c2 = idx / (ND * 32), rem = idx - c2 * (ND * 32), dg = rem >> 5, lr = rem & 31;
        const uint4 w = *reinterpret_cast<const uint4 *>(img + c2 * CS + dg * 512 + ((lr ^ c2) << 4));
        *reinterpret_cast<uint4 *>(a.A + ((((((size_t)m * N + x0 + c2) * a.nch + ch) * 2 + rt) * ND + dg) * 1024) + khalf * 512 + lr * 16) = w;
    }
}
// ---- panel words -> B.  workgroup = (modulus m, column tile jt, 16 k, 32 coefficients): 256 (k, column) rows of 256 contiguous bytes in, 160 pieces of 256
// contiguous bytes out (the 16-k quarter of a 1 KiB tile).  A thread takes four consecutive k of one (column, coefficient), so a digit of the four is one dword.
constexpr int I8_PP = 32;                                   // coefficients per workgroup
template <int ND>
__global__ void __launch_bounds__(256) k_i8_pack_pt(I8Args a) {
    constexpr int PSTR = ND * 256 + 4;                      // bytes per coefficient in the image (+ 4: lanes = coefficients fall on distinct banks)
    __shared__ __attribute__((aligned(16))) unsigned char img[I8_PP * PSTR];       // [cc 32][b ND][j 16][kk 16]
    const int H = SFG_N / 2, tid = threadIdx.x;
    int b = blockIdx.x;
    const int cb = b % (H / I8_PP); b /= H / I8_PP;
    const int kq = b % (a.nch * 4); b /= a.nch * 4;
    const int jt = b % a.njt, m = b / a.njt;
    const int c0 = cb * I8_PP, cc = tid & (I8_PP - 1), slot = tid >> 5;
    const u64 *src = a.pt + a.pt_l0_off + (size_t)m * a.pt_l_stride + c0 + cc;
#pragma unroll 2
    for (int it = 0; it < 8; it++) {
        const int item = it * 8 + slot, j = item >> 2, k4 = item & 3, n = jt * 16 + j;
        unsigned dw[ND];
#pragma unroll
        for (int i = 0; i < ND; i++) dw[i] = 0;
#pragma unroll
        for (int x = 0; x < 4; x++) {
            const int k = kq * 16 + k4 * 4 + x;
            long long v = 0;
            if (k < a.K && n < a.Ncols) {
                const u64 w = src[(size_t)n * a.pt_n_stride + (size_t)k * a.pt_k_stride];
                v = ND == 6 ? (long long)w : (long long)((w & 0xFFFULL) | (((w >> 16) & 0xFFFULL) << 12) | ((w >> 32) << 24));       // plain word / packed-limb panel word (pack_limbs)
            }
            int8_t d[ND]; i8_digits<ND>(v, d);
#pragma unroll
            for (int i = 0; i < ND; i++) dw[i] |= (unsigned)(uint8_t)d[i] << (8 * x);
        }
#pragma unroll
        for (int i = 0; i < ND; i++) *reinterpret_cast<unsigned *>(img + cc * PSTR + i * 256 + j * 16 + k4 * 4) = dw[i];
    }
    __syncthreads();
    // piece (cc, digit): 256 bytes = 16 lanes x 16 bytes, at byte (kq % 4) * 256 of its tile
    const int ch = kq >> 2, g = kq & 3, l16 = tid & 15;
    for (int pc = tid >> 4; pc < I8_PP * ND; pc += 16) {
        const int c2 = pc / ND, d = pc % ND;
        const unsigned *sp = reinterpret_cast<const unsigned *>(img + c2 * PSTR + d * 256 + l16 * 16);
        const uint4 w = make_uint4(sp[0], sp[1], sp[2], sp[3]);
        *reinterpret_cast<uint4 *>(a.B + ((((((size_t)m * H + c0 + c2) * a.njt + jt) * a.nch + ch) * ND + d) * 1024) + g * 256 + l16 * 16) = w;
    }
}
// ---- the same from digit planes (the plaintext NTT's output when the int8 MAC is on: five planes of N/2 bytes in a row's 64 KiB).  workgroup = (modulus m,
// column tile jt, 16 k, 128 coefficients), one digit at a time: 256 (k, column) rows of 128 contiguous bytes in; a lane takes four consecutive k of FOUR consecutive
// coefficients (a dword each) and a 4 x 4 byte transpose turns them into one dword of four k per coefficient.  Image [cc 128][j 16][k4 4] dwords, j and k4 XORed with
// bits of the lane's coefficient group so that the 32 lanes of a row group hit 32 banks.
template <int ND>
__global__ void __launch_bounds__(256) k_i8_pack_pt_digits(I8Args a) {
    __shared__ __attribute__((aligned(16))) unsigned img[I8_PD * 64];
    const int H = SFG_N / 2, tid = threadIdx.x;
    int b = blockIdx.x;
    const int cb = b % (H / I8_PD); b /= H / I8_PD;
    const int kq = b % (a.nch * 4); b /= a.nch * 4;
    const int jt = b % a.njt, m = b / a.njt;
    const int c0 = cb * I8_PD, cq = tid & 31, slot = tid >> 5;
    const int ch = kq >> 2, g = kq & 3, l16 = tid & 15;
    const unsigned char *src = reinterpret_cast<const unsigned char *>(a.pt + a.pt_l0_off + (size_t)m * a.pt_l_stride) + (size_t)cb * a.pt_cb_stride + cq * 4;
    constexpr int NI = 8;                                   // items whose loads are issued together (4 dwords each; 2, 4, 8 measured: 8 = the plain loop's time, round 4)
    for (int d = 0; d < ND; d++) {
        for (int it0 = 0; it0 < 8; it0 += NI) {
            unsigned w[NI][4];
#pragma unroll
            for (int u = 0; u < NI; u++) {
                const int item = (it0 + u) * 8 + slot, j = item >> 2, k4 = item & 3, n = jt * 16 + j;
#pragma unroll
                for (int x = 0; x < 4; x++) {
                    const int k = kq * 16 + k4 * 4 + x;
                    const bool ok = k < a.K && n < a.Ncols;
                    const unsigned v = *reinterpret_cast<const unsigned *>(src + (ok ? ((size_t)n * a.pt_n_stride + (size_t)k * a.pt_k_stride) * 8 : (size_t)0) + (size_t)d * a.pt_d_stride);
                    w[u][x] = ok ? v : 0u;
                }
            }
#pragma unroll
            for (int u = 0; u < NI; u++) {
                const int item = (it0 + u) * 8 + slot, j = item >> 2, k4 = item & 3;
                unsigned o[4]; bytes_tr4(w[u][0], w[u][1], w[u][2], w[u][3], o);
#pragma unroll
                for (int e = 0; e < 4; e++) img[(cq * 4 + e) * 64 + ((j ^ (cq & 7)) << 2) + (k4 ^ (cq >> 3))] = o[e];
            }
        }
        __syncthreads();
        for (int pc = tid >> 4; pc < I8_PD; pc += 16) {
            const int q2 = pc >> 2;
            const unsigned *sp = img + pc * 64 + ((l16 ^ (q2 & 7)) << 2);
            const uint4 w = make_uint4(sp[0 ^ (q2 >> 3)], sp[1 ^ (q2 >> 3)], sp[2 ^ (q2 >> 3)], sp[3 ^ (q2 >> 3)]);
            *reinterpret_cast<uint4 *>(a.B + ((((((size_t)m * H + c0 + pc) * a.njt + jt) * a.nch + ch) * ND + d) * 1024) + g * 256 + l16 * 16) = w;
        }
        __syncthreads();
    }
}
// ---- the same as a low-occupancy mover (i8_move.hpp): job.nblocks workgroups (one per CU) walk the items with the next units' loads in flight in registers.
// Its place is in front of the plaintext NTT's workgroups (k_ntt_half3_move, ntt.hip: the riding transposition); alone on the chip it moves what no NTT launch took
// (i8_ride_finish) and, in the A/B build, is the A/B of the pass above.
template <int DEPTH, bool NT>
__global__ void __launch_bounds__(256, 4) k_i8_move_pt(MoveJob job) {
    __shared__ __attribute__((aligned(16))) unsigned img[I8_PD * 64];
    i8_move_block<DEPTH, NT>(job, blockIdx.x, img, (int)threadIdx.x);
}
#ifdef SFG_AB          // streamed transposition (round 4 - 5 experiment, measured slower) and the mover form of the pass (round 6, no gain): A/B build only
// ---- the same from the DENSE digit planes of one encode batch (StagePack, kernels.hpp): plaintext p of the batch is shift shift0 + p = giant n, baby b of block
// row g; its byte goes to column n, k' = g * kb + b.  A workgroup = (modulus, column tile jt, 16 k', 128 coefficients) as above, restricted to what this batch
// owns: columns [n_lo, n_hi) x the dwords of block row g (kb is a multiple of 4, so a 16-byte run splits between block rows on dword boundaries).  Owned positions
// without a plaintext - the pad baby 91, shifts past 8191 - are written as zeros; what no batch owns (columns 91..95, k' past the last block row) is zeroed when the
// tile buffer is (re)shaped.  Reads come from the staging buffer the NTT has just written (Infinity Cache), writes are 16-byte pieces of 1 KiB tiles.
struct I8StageArgs { const u64 *stage; int8_t *B; int L, l0, nl, shift0, nshift, n_lo, n_hi, g, kb, njt, nch, jt0, njt_b, kq0, nkq; };
template <int ND>
__global__ void __launch_bounds__(256) k_i8_pack_stage(I8StageArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned img[I8_PD * 64];
    const int H = SFG_N / 2, tid = threadIdx.x;
    int b = blockIdx.x;
    const int cb = b % (H / I8_PD); b /= H / I8_PD;
    const int kq = a.kq0 + b % a.nkq; b /= a.nkq;
    const int jt = a.jt0 + b % a.njt_b, m = b / a.njt_b;
    const int c0 = cb * I8_PD, cq = tid & 31, slot = tid >> 5;
    const int ch = kq >> 2, g16 = kq & 3, l16 = tid & 15;
    const unsigned char *src = reinterpret_cast<const unsigned char *>(a.stage) + (size_t)(a.l0 + m) * H * 8 + c0 + cq * 4;
    const size_t pstride = (size_t)a.L * H * 8;                       // bytes per plaintext of the staging buffer
    // ownership of this workgroup's stores: lane l16 = column n, dword k4 of a 16-byte piece = k' in [kq*16 + 4 k4, + 4)
    const int n_st = jt * 16 + l16; const bool lane_owned = n_st >= a.n_lo && n_st < a.n_hi;
    unsigned own = 0;
#pragma unroll
    for (int k4 = 0; k4 < 4; k4++) { const int k0 = kq * 16 + k4 * 4; if (k0 >= a.g * a.kb && k0 < (a.g + 1) * a.kb) own |= 1u << k4; }
    for (int d = 0; d < ND; d++) {
#pragma unroll 2
        for (int it = 0; it < 8; it++) {
            const int item = it * 8 + slot, j = item >> 2, k4 = item & 3, n = jt * 16 + j;
            unsigned w[4];
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const int k = kq * 16 + k4 * 4 + x, baby = k - a.g * a.kb, p = n * SFG_D + baby - a.shift0;
                const bool ok = baby >= 0 && baby < SFG_D && n >= a.n_lo && n < a.n_hi && p >= 0 && p < a.nshift;
                w[x] = ok ? *reinterpret_cast<const unsigned *>(src + (size_t)p * pstride + (size_t)d * H) : 0u;
            }
            unsigned o[4]; bytes_tr4(w[0], w[1], w[2], w[3], o);
#pragma unroll
            for (int e = 0; e < 4; e++) img[(cq * 4 + e) * 64 + ((j ^ (cq & 7)) << 2) + (k4 ^ (cq >> 3))] = o[e];
        }
        __syncthreads();
        if (lane_owned && own)
            for (int pc = tid >> 4; pc < I8_PD; pc += 16) {
                const int q2 = pc >> 2;
                const unsigned *sp = img + pc * 64 + ((l16 ^ (q2 & 7)) << 2);
                const uint4 w = make_uint4(sp[0 ^ (q2 >> 3)], sp[1 ^ (q2 >> 3)], sp[2 ^ (q2 >> 3)], sp[3 ^ (q2 >> 3)]);
                unsigned *dst = reinterpret_cast<unsigned *>(a.B + ((((((size_t)m * H + c0 + pc) * a.njt + jt) * a.nch + ch) * ND + d) * 1024) + g16 * 256 + l16 * 16);
                if (own == 15u) *reinterpret_cast<uint4 *>(dst) = w;
                else { if (own & 1u) dst[0] = w.x; if (own & 2u) dst[1] = w.y; if (own & 4u) dst[2] = w.z; if (own & 8u) dst[3] = w.w; }
            }
        __syncthreads();
    }
}
#endif
// ---- the MAC.  grid = nl * N/2 workgroups of njt waves
template <int ND>
__global__ void __launch_bounds__(384, 1) k_mac_i8(I8Args a, const ModConst *modc) {
    const int N = SFG_N, H = N / 2;
    const int lane = threadIdx.x & 63;
    // wave -> (modulus, coefficient pair, column tile).  Default: the njt column waves of a pair are one workgroup (blockDim = 64 njt); the diagnostic launch
    // SFG_MAC_I8_WG=1 gives every wave its own workgroup (consecutive workgroups go to different XCDs: no cache shared between the column waves of a pair)
    const int gw = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int jt = gw % a.njt, pr = gw / a.njt;
    const int c = pr % H, m = pr / H;
    const double q = modc[a.l0 + m].q, qinv = modc[a.l0 + m].qinv;
    v4i acc[4][2 * ND - 1];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int s = 0; s < 2 * ND - 1; s++) acc[t][s] = (v4i){0, 0, 0, 0};
    const uint4 *Bp = reinterpret_cast<const uint4 *>(a.B) + ((((size_t)m * H + c) * a.njt + jt) * a.nch) * ND * 64 + lane;
    const uint4 *A0 = reinterpret_cast<const uint4 *>(a.A) + (((size_t)m * N + c) * a.nch) * 2 * ND * 64 + lane;
    const uint4 *A1 = reinterpret_cast<const uint4 *>(a.A) + (((size_t)m * N + (N - 1 - c)) * a.nch) * 2 * ND * 64 + lane;
#pragma unroll 1
    for (int ch = 0; ch < a.nch; ch++) {
        v4i b[ND];
#pragma unroll
        for (int d = 0; d < ND; d++) b[d] = __builtin_nontemporal_load(reinterpret_cast<const v4i *>(Bp + (size_t)(ch * ND + d) * 64));       // read once: streamed past the caches, which keep the rot tiles the six waves share
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const uint4 *Ap = (t < 2 ? A0 : A1) + (size_t)((ch * 2 + (t & 1)) * ND) * 64;
#pragma unroll
            for (int x = 0; x < ND; x++) {
                const uint4 w = Ap[(size_t)x * 64];
                const v4i av = (v4i){(int)w.x, (int)w.y, (int)w.z, (int)w.w};
#pragma unroll
                for (int d = 0; d < ND; d++) acc[t][x + d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, b[d], acc[t][x + d], 0, 0, 0);
            }
        }
        // (no workgroup barrier: a barrier drains every wave's loads at every chunk and exposes the load latency 23 times per workgroup; left alone the six waves
        //  drift by a few chunks and still find the pair's rot tiles in the L2)
    }
    // sum_s D_s 256^s mod q by Horner (i8_horner), canonical
    const bool wide = q > I8_WIDE_Q;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        u64 *o = a.T + ((((((size_t)m * H + c) * 2 + (t >> 1)) * a.njt + jt) * 2 + (t & 1)) * 64 + lane) * 4;
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = (u64)i8_horner(acc[t], e, q, qinv, wide);
    }
}
#ifdef SFG_AB          // LDS-staged rot tiles (round 3, measured slower than the cache-shared and the ring forms): A/B build only
// ---- the same with the rot tiles of the pair staged through LDS (six column waves: the product's 91 columns).  Through the cache alone the six waves fetched
// them 2.8 x (PMC); here the workgroup loads the 20 KiB of a chunk once - the next chunk's pieces travel in registers beside the current chunk's MFMAs - and every
// wave reads its 20 operand tiles from the 2 x 20 KiB image.
__global__ void __launch_bounds__(384, 1) k_mac_i8_lds(I8Args a, const ModConst *modc) {
    __shared__ uint4 As[2][2 * 2 * I8_ND * 64];
    const int N = SFG_N, H = N / 2, tid = threadIdx.x;
    const int lane = tid & 63, jt = tid >> 6;
    const int c = blockIdx.x % H, m = blockIdx.x / H;
    const double q = modc[a.l0 + m].q, qinv = modc[a.l0 + m].qinv;
    v4i acc[4][9];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int s = 0; s < 9; s++) acc[t][s] = (v4i){0, 0, 0, 0};
    const uint4 *Bp = reinterpret_cast<const uint4 *>(a.B) + ((((size_t)m * H + c) * a.njt + jt) * a.nch) * I8_ND * 64 + lane;
    const uint4 *A0 = reinterpret_cast<const uint4 *>(a.A) + (((size_t)m * N + c) * a.nch) * 2 * I8_ND * 64;
    const uint4 *A1 = reinterpret_cast<const uint4 *>(a.A) + (((size_t)m * N + (N - 1 - c)) * a.nch) * 2 * I8_ND * 64;
    constexpr int HALF = 2 * I8_ND * 64;               // uint4 per coefficient and chunk (10 KiB)
    // piece i < 2 HALF of chunk ch: coefficient half i / HALF, offset i % HALF; thread tid takes pieces tid, tid + 384, ...
    auto src = [&](int ch, int i) { return (i < HALF ? A0 : A1) + (size_t)ch * HALF + (i < HALF ? i : i - HALF); };
    uint4 stage[4];
#pragma unroll
    for (int u = 0; u < 4; u++) { const int i = tid + 384 * u; if (i < 2 * HALF) As[0][i] = *src(0, i); }
    __syncthreads();
#pragma unroll 1
    for (int ch = 0; ch < a.nch; ch++) {
        const bool more = ch + 1 < a.nch;
        if (more) {
#pragma unroll
            for (int u = 0; u < 4; u++) { const int i = tid + 384 * u; if (i < 2 * HALF) stage[u] = *src(ch + 1, i); }
        }
        v4i b[I8_ND];
#pragma unroll
        for (int d = 0; d < I8_ND; d++) { const uint4 w = Bp[(size_t)(ch * I8_ND + d) * 64]; b[d] = (v4i){(int)w.x, (int)w.y, (int)w.z, (int)w.w}; }
        const uint4 *Ac = As[ch & 1];
#pragma unroll
        for (int t = 0; t < 4; t++) {
#pragma unroll
            for (int x = 0; x < I8_ND; x++) {
                const uint4 w = Ac[(t * I8_ND + x) * 64 + lane];
                const v4i av = (v4i){(int)w.x, (int)w.y, (int)w.z, (int)w.w};
#pragma unroll
                for (int d = 0; d < I8_ND; d++) acc[t][x + d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, b[d], acc[t][x + d], 0, 0, 0);
            }
        }
        if (more) {
#pragma unroll
            for (int u = 0; u < 4; u++) { const int i = tid + 384 * u; if (i < 2 * HALF) As[(ch + 1) & 1][i] = stage[u]; }
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
        u64 *o = a.T + ((((((size_t)m * H + c) * 2 + (t >> 1)) * a.njt + jt) * 2 + (t & 1)) * 64 + lane) * 4;
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = (u64)i8_horner(acc[t], e, q, qinv, false);       // (five digits: q < 2^39)
    }
}
#endif
// ---- the MAC with both operand streams prefetched through an LDS ring by the DMA engine (round 4; the default for full 91-column launches).
// Counters (profiles/r04_pmc_mac_i8.json) show that k_mac_i8 fetches exactly its operand bytes (31.88 GB per launch against 31.88 GB algorithmic: the six column
// waves of a pair do share the rot tiles, in the L1) - and yet runs at 55 % of the achievable HBM rate: a wave requests a chunk's 25 KiB, waits for all of it,
// then issues 100 MFMAs with nothing in flight; the round time is memory time PLUS matrix time.  Here a workgroup's chunk (20 KiB of rot tiles, loaded ONCE,
// + 6 x 5 KiB of plaintext tiles) travels global -> LDS by global_load_lds_dwordx4 two chunks ahead of its use (three 50 KiB slots, counted vmcnt, one raw
// s_barrier per chunk), needs no VGPRs on the way, and the MFMA operands come from LDS by ds_read_b128 issued three tiles ahead (counted lgkmcnt, by hand:
// left to the compiler, every LDS read after a DMA instruction is preceded by s_waitcnt vmcnt(0)).  Identical arithmetic, identical words.
__device__ __forceinline__ void i8_dma16(const void *gsrc, void *lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc, (__attribute__((address_space(3))) void *)lds_base, 16, 0, 0);
}
// DEPTH slots of 10 ND KiB: three for the 35-bit moduli (150 KiB), two for the 46-bit one (120 KiB: its 144 MFMAs per chunk cover one chunk of lookahead)
// DIAG (timing diagnostics only, results invalid; A/B build): 1 = one MFMA per rot tile instead of ND (the matrix pipe nearly idle), 2 = no DMA after the prologue
// (the memory system idle): which side of the ring sets the chunk time
// HALVES = 2: twelve waves, a wave = one COEFFICIENT (c or N-1-c) x 16 columns: 2 row tiles x (2 ND - 1) sums, three waves on every SIMD instead of 2-2-1-1.  The
// timing diagnostics showed the matrix side of the six-wave form (3.85 ms of a 5.02 ms launch with the memory system idle; 4.29 ms with the matrix pipe idle) set by
// the two doubly occupied SIMDs; with LDS-staged operands the second wave of a pair costs no extra fetch.
template <int ND, int I8R_DEPTH, int DIAG = 0, int HALVES = 1>
__global__ void __launch_bounds__(384 * HALVES, 1) k_mac_i8_ring(I8Args a, const ModConst *modc) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ring[];
    constexpr int NA = 4 * ND, NB = 6 * ND, SLOT = (NA + NB) * 1024;                  // tiles per slot
    constexpr int NT = 4 / HALVES, NAW = NT * ND;                                       // row tiles / rot tiles of one wave
    // DMA instructions per wave and chunk.  Six waves: own plaintext tiles + every sixth rot tile.  Twelve: the waves of coefficient c fetch the plaintext tiles of
    // their column, the waves of N-1-c the rot tiles (every sixth; the last round wraps around onto tiles already fetched - same bytes to the same place)
    constexpr int AR = (NA + 5) / 6, NJ0 = HALVES == 1 ? ND + AR : ND, NJ1 = HALVES == 1 ? ND + AR : AR;
    static_assert(NJ0 < 32 && NJ1 < 32 && SLOT * I8R_DEPTH <= 160 * 1024 && SLOT < 65536, "ring budget");
    const int N = SFG_N, H = N / 2;
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), jt = wv % 6, hf = wv / 6;      // (scalar: wave-uniform branches below) column tile; (HALVES = 2) coefficient of the pair
    const int c = blockIdx.x % H, m = blockIdx.x / H, nch = a.nch;
    v4i acc[NT][2 * ND - 1];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int s = 0; s < 2 * ND - 1; s++) acc[t][s] = (v4i){0, 0, 0, 0};
    const unsigned char *gB = reinterpret_cast<const unsigned char *>(a.B) + ((((size_t)m * H + c) * 6 + jt) * nch) * ND * 1024 + lane * 16;
    const unsigned char *gA0 = reinterpret_cast<const unsigned char *>(a.A) + (((size_t)m * N + c) * nch) * 2 * ND * 1024 + lane * 16;
    const unsigned char *gA1 = reinterpret_cast<const unsigned char *>(a.A) + (((size_t)m * N + (N - 1 - c)) * nch) * 2 * ND * 1024 + lane * 16;
    // the rot tiles of a chunk: tile j = (coefficient, row tile, digit) = t * ND + x of the MFMA loop below
    auto issue = [&](int ch, int slot) {
        unsigned char *sl = ring + slot * SLOT;
        if (HALVES == 1 || hf == 0) {
#pragma unroll
            for (int d = 0; d < ND; d++) i8_dma16(gB + ((size_t)ch * ND + d) * 1024, sl + (NA + jt * ND + d) * 1024);
        }
        if (HALVES == 1 || hf == 1) {
#pragma unroll
            for (int r = 0; r < AR; r++) {
                int j = r * 6 + jt; if (j >= NA) j -= NA;
                const int half = j / (2 * ND), rem = j - half * 2 * ND;
                i8_dma16((half ? gA1 : gA0) + ((size_t)ch * 2 * ND + rem) * 1024, sl + j * 1024);
            }
        }
    };
#pragma unroll
    for (int ch = 0; ch < I8R_DEPTH - 1; ch++) if (ch < nch) issue(ch, ch);
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)ring + (unsigned)lane * 16u;
    int slot = 0;
#pragma unroll 1
    for (int ch = 0; ch < nch; ch++) {
        // chunk ch has landed once every wave's own pieces have (the chunk issued after it may still be in flight) and the workgroup has met; the meeting also
        // says that everybody has finished reading chunk ch - 1, whose slot is refilled next
        if (DIAG == 2 || ch + I8R_DEPTH - 2 >= nch) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (HALVES == 1 || hf == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NJ0 * (I8R_DEPTH - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NJ1 * (I8R_DEPTH - 2)) : "memory");
        __builtin_amdgcn_s_barrier();
        if (DIAG != 2 && ch + I8R_DEPTH - 1 < nch) issue(ch + I8R_DEPTH - 1, slot == 0 ? I8R_DEPTH - 1 : slot - 1);
        const unsigned ab = lds0 + (unsigned)(slot * SLOT) + (unsigned)(hf * NAW * 1024), bb = lds0 + (unsigned)(slot * SLOT) + (unsigned)((NA + jt * ND) * 1024);
        v4i b[ND], ar[5];
#pragma unroll
        for (int d = 0; d < ND; d++) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(b[d]) : "v"(bb), "n"(d * 1024) : "memory");
#pragma unroll
        for (int i = 0; i < 3; i++) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(ar[i]) : "v"(ab), "n"(i * 1024) : "memory");
#pragma unroll
        for (int i = 0; i < NAW; i++) {
            // rot tile i is in: the LDS unit answers in order, and at most min(2, NAW - 1 - i) reads were issued after it
            if (i == 0) {
                if constexpr (ND == 5) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ar[0]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4])::"memory");
                else asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ar[0]), "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[ND - 1])::"memory");
            } else if (i < NAW - 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ar[i % 5])::"memory");
            else if (i == NAW - 2) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(ar[i % 5])::"memory");
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ar[i % 5])::"memory");
            // tile i + 3 goes into the registers of tile i - 2 (tile i is named as an operand so that its MFMAs stay BEHIND this request: three tiles in flight)
            if (i + 3 < NAW) asm volatile("ds_read_b128 %0, %2 offset:%3" : "=&v"(ar[(i + 3) % 5]), "+v"(ar[i % 5]) : "v"(ab), "n"((i + 3) * 1024) : "memory");
            const int t = i / ND, x = i - t * ND;
#pragma unroll
            for (int d = 0; d < (DIAG == 1 ? 1 : ND); d++) acc[t][x + d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(ar[i % 5], b[d], acc[t][x + d], 0, 0, 0);
        }
        slot = slot + 1 == I8R_DEPTH ? 0 : slot + 1;
    }
    const double q = modc[a.l0 + m].q, qinv = modc[a.l0 + m].qinv;
    const bool wide = ND == 6 && q > I8_WIDE_Q;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int tg = hf * NT + t;                                                     // tile of the pair: (coefficient, row tile)
        u64 *o = a.T + ((((((size_t)m * H + c) * 2 + (tg >> 1)) * 6 + jt) * 2 + (tg & 1)) * 64 + lane) * 4;
#pragma unroll
        for (int e = 0; e < 4; e++) o[e] = (u64)i8_horner(acc[t], e, q, qinv, wide);
    }
}
// ---- tile-ordered results -> canonical accumulators.  workgroup = (m, 16 coefficient pairs, half, jt, rt): 256 (n, r) rows x 16 coefficients through LDS
__global__ void __launch_bounds__(256) k_i8_untile(I8Args a, const ModConst *modc) {
    __shared__ u64 img[16][257];
    const int N = SFG_N, H = N / 2, tid = threadIdx.x;
    int b = blockIdx.x;
    const int rt = b & 1; b >>= 1;
    const int jt = b % a.njt; b /= a.njt;
    const int half = b & 1; b >>= 1;
    const int cb = b % (H / 16), m = b / (H / 16);
    const double q = modc[a.l0 + m].q;
    for (int cc = 0; cc < 16; cc++) {
        const u64 *src = a.T + (((((size_t)m * H + cb * 16 + cc) * 2 + half) * a.njt + jt) * 2 + rt) * 256;
        img[cc][tid] = src[tid];                                       // tid = lane * 4 + e
    }
    __syncthreads();
    // element (lane, e) of a tile: column j = lane & 15, row i = (lane >> 4) * 4 + e
    const int cc = tid & 15;
    const int x = half ? N - 1 - (cb * 16 + cc) : cb * 16 + cc;
    for (int p = tid >> 4; p < 256; p += 16) {
        const int lane = p >> 2, e = p & 3, n = jt * 16 + (lane & 15), r = a.r0 + rt * 16 + (lane >> 4) * 4 + e;
        if (n >= a.Ncols || r >= a.R) continue;
        u64 *o = a.out + (size_t)n * a.out_n_stride + (size_t)r * a.out_r_stride + (size_t)(a.l0 + m) * N + x;
        double v = (double)img[cc][p];
        if (a.accumulate) { v += (double)*o; if (v >= q) v -= q; }
        *o = (u64)v;
    }
}

int mac_i8_set_attrs(sfg_ctx *ctx) {       // per device, at context creation (ctx.hip)
    SFG_HIP(ctx, hipFuncSetAttribute((const void *)k_i8_pack_rot<5>, hipFuncAttributeMaxDynamicSharedMemorySize, I8_PC * i8_rot_img_stride<5>()));
    SFG_HIP(ctx, hipFuncSetAttribute((const void *)k_i8_pack_rot<6>, hipFuncAttributeMaxDynamicSharedMemorySize, I8_PC * i8_rot_img_stride<6>()));
    SFG_HIP(ctx, hipFuncSetAttribute((const void *)k_mac_i8_ring<5, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 10 * 5 * 1024));
    SFG_HIP(ctx, hipFuncSetAttribute((const void *)k_mac_i8_ring<6, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 10 * 6 * 1024));
    SFG_HIP(ctx, hipFuncSetAttribute((const void *)k_mac_i8_ring<5, 3, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 10 * 5 * 1024));
    SFG_HIP(ctx, hipFuncSetAttribute((const void *)k_mac_i8_ring<6, 2, 0, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 10 * 6 * 1024));
#ifdef SFG_AB
    SFG_HIP(ctx, hipFuncSetAttribute((const void *)k_mac_i8_ring<5, 3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 10 * 5 * 1024));
    SFG_HIP(ctx, hipFuncSetAttribute((const void *)k_mac_i8_ring<5, 3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * 10 * 5 * 1024));
#endif
    return 0;
}
// bytes of the two operand streams and the tile-ordered results of one launch (for the group-size choice in matmul.hip)
size_t mac_i8_stream_bytes(int K, int nl, int ND, int copies_of_rot) {
    const size_t N = SFG_N, H = N / 2, nch = ((size_t)K + 63) / 64;
    return (size_t)nl * (N * nch * 2 * ND * 1024 * copies_of_rot + H * 6 * nch * ND * 1024 + H * 2 * 6 * 2 * 256 * 8);
}
int launch_i8_pack_stage(sfg_ctx *ctx, StagePack &sp, int shift_lo, int nshift, int L) {
#ifndef SFG_AB
    (void)sp; (void)shift_lo; (void)nshift; (void)L;
    SFG_FAIL(ctx, "the streamed transposition exists in the A/B build only (make ab)");
#else
    const int H = SFG_N / 2, d = SFG_D;
    if (shift_lo % d) SFG_FAIL(ctx, "i8 stage pack: internal: a batch starts inside a giant step");
    I8StageArgs a;
    a.stage = sp.stage; a.L = L; a.shift0 = shift_lo; a.nshift = nshift; a.g = sp.g; a.kb = sp.kb; a.njt = sp.njt; a.nch = sp.nch;
    a.n_lo = shift_lo / d; a.n_hi = (shift_lo + nshift + d - 1) / d;
    if (shift_lo + nshift >= SFG_SLOTS) a.n_hi = d;                  // the last batch owns all of giant 90 (its shifts past 8191 are zero plaintexts)
    a.jt0 = a.n_lo / 16; a.njt_b = (a.n_hi - 1) / 16 - a.jt0 + 1;
    a.kq0 = (sp.g * sp.kb) / 16; a.nkq = ((sp.g + 1) * sp.kb - 1) / 16 - a.kq0 + 1;
    hipStream_t saved = ctx->stream; ctx->stream = sp.q;
    const bool sampled = (sp.seq++ & 7) == 0;
    {
        PhaseTimer t(ctx, "mac_i8_pack_pt", sampled);
        if (sp.n_small) {
            a.B = sp.Bs; a.l0 = sp.l_small0; a.nl = sp.n_small;
            hipLaunchKernelGGL(k_i8_pack_stage<5>, dim3((unsigned)((size_t)a.nl * a.njt_b * a.nkq * (H / I8_PD))), dim3(256), 0, sp.q, a);
        }
        if (sp.l_big >= 0) {
            a.B = sp.Bb; a.l0 = sp.l_big; a.nl = 1;
            hipLaunchKernelGGL(k_i8_pack_stage<6>, dim3((unsigned)((size_t)a.njt_b * a.nkq * (H / I8_PD))), dim3(256), 0, sp.q, a);
        }
        if (sampled) t.stop(8, 8.0 * nshift * H * (5.0 * sp.n_small + (sp.l_big >= 0 ? 6.0 : 0.0)) * 2.0);       // (one launch pair in eight is timed: counted for eight)
    }
    ctx->stream = saved;
    SFG_HIP(ctx, hipGetLastError());
    return 0;
#endif
}
// bytes of the tile buffer of `nl` moduli with ND digits for K' contraction steps
size_t mac_i8_tile_bytes(int Kp, int nl, int ND) { return (size_t)nl * (SFG_N / 2) * 6 * (((size_t)Kp + 63) / 64) * ND * 1024; }
static void launch_move_alone(hipStream_t q, const MoveJob &j) {
#define SFG_MV(D, T) hipLaunchKernelGGL((k_i8_move_pt<D, T>), dim3(j.nblocks), dim3(256), 0, q, j)
#ifdef SFG_AB
    if (j.depth == 3) { if (j.nt) SFG_MV(3, true); else SFG_MV(3, false); return; }
    if (j.depth == 2) { if (j.nt) SFG_MV(2, true); else SFG_MV(2, false); return; }
    if (!j.nt) { SFG_MV(1, false); return; }
#endif
    SFG_MV(1, true);
#undef SFG_MV
}
// ---- the riding transposition (kernels.hpp PtRide): tile buffers, the job of one delayed MAC launch, what is left of it
static bool i8_ride_moduli(sfg_ctx *ctx, int L, int &l_small0, int &n_small, int &l_big) {
    std::vector<int> plane_of, is_big; if (mac_dma_planes(ctx, L, plane_of, is_big) < 0) return false;
    l_big = -1; l_small0 = -1; n_small = 0; int nbig = 0;
    for (int l = 0; l < L; l++) { if (is_big[l]) { l_big = l; nbig++; } else { if (l_small0 < 0) l_small0 = l; n_small++; } }
    if (nbig > 1 || !n_small) return false;
    for (int l = 0; l < L; l++) if (!is_big[l] && (l < l_small0 || l >= l_small0 + n_small)) return false;      // the 35-bit moduli must be one run: one MAC launch, one tile buffer
    if (l_big >= 0 && (!ctx->cfg.mac_i8_big || ctx->q[l_big] > SFG_I8_BIG_QMAX)) return false;
    return true;
}
int i8_ride_tiles(sfg_ctx *ctx, int K, int L, int8_t **Bs, int8_t **Bb) {
    int l_small0, n_small, l_big;
    *Bs = *Bb = nullptr;
    if (!i8_ride_moduli(ctx, L, l_small0, n_small, l_big)) return 0;
    SFG_TRY(sfg_scratch(ctx, "mi8.Bs", mac_i8_tile_bytes(K, n_small, 5), (void **)Bs));
    if (l_big >= 0) SFG_TRY(sfg_scratch(ctx, "mi8.Bb", mac_i8_tile_bytes(K, 1, 6), (void **)Bb));
    return 0;
}
// Panel layouts (MacStrides::pt_layout).  0: L rows of N/2 words per plaintext; 1 (compact): a plaintext's 5 / 6 digit planes of N/2 bytes per modulus back to back;
// 2 (K-major): [column][plane][128-byte coefficient block][k < K][128 B].  Fills the row addressing of modulus l0 (and the like moduli after it) in `a`; pt_k / pt_n
// (words between consecutive k / columns) are the caller's for layouts 0 and 1 and follow from K for layout 2.
void i8_panel_rows(const sfg_ctx *ctx, int layout, int K, int L, int l, bool big, I8Args &a) {
    const size_t H = SFG_N / 2;
    a.pt_d_stride = H; a.pt_cb_stride = I8_PD;
    if (layout == 0) { a.pt_l0_off = (size_t)l * H; a.pt_l_stride = H; return; }
    size_t below = 0, all = 0; for (int t = 0; t < L; t++) { const size_t pl = ctx->q[t] < (1ULL << 36) ? 5 : 6; if (t < l) below += pl; all += pl; }
    const size_t nd = big ? 6 : 5;
    if (layout == 1) { a.pt_l0_off = below * H / 8; a.pt_l_stride = nd * H / 8; return; }
    const size_t blk = (size_t)K * 128;                          // bytes of one coefficient block of a plane: K rows of 128 B
    a.pt_d_stride = 64 * blk; a.pt_cb_stride = blk;
    a.pt_l0_off = below * 64 * blk / 8; a.pt_l_stride = nd * 64 * blk / 8;
    a.pt_k_stride = 16; a.pt_n_stride = all * 64 * blk / 8;
}
int i8_ride_prepare(sfg_ctx *ctx, const u64 *panel, int K, int Ncols, size_t pt_k, size_t pt_n, int layout, int L, int launches, PtRide &ride) {
    const int H = SFG_N / 2;
    ride = PtRide();
    int l_small0, n_small, l_big;
    if (ctx->cfg.pt_ride <= 0 || launches < 1 || Ncols > 96 || !i8_ride_moduli(ctx, L, l_small0, n_small, l_big)) return 0;
    if (layout != 2 && (pt_n * 8 >= (1ULL << 31) || pt_k * 8 * 16 >= (1ULL << 31))) return 0;                  // the mover's per-lane offsets are 32-bit
    if (layout == 2 && (size_t)K * 128 * 64 * 32 >= (1ULL << 31)) return 0;                   // (K-major: a column is at most 32 planes of 64 blocks of K x 128 B)
    int8_t *Bs, *Bb; SFG_TRY(i8_ride_tiles(ctx, K, L, &Bs, &Bb));
    const int nch = (K + 63) / 64, njt = (Ncols + 15) / 16;
    auto fill = [&](I8Args &a, int l0, int nl, int8_t *B) {
        memset(&a, 0, sizeof a);
        a.pt = panel; a.pt_k_stride = pt_k; a.pt_n_stride = pt_n; a.K = K; a.Ncols = Ncols; a.l0 = l0; a.nl = nl; a.nch = nch; a.njt = njt; a.pt_digits = 1; a.B = B;
        i8_panel_rows(ctx, layout, K, L, l0, l0 == l_big, a);
    };
    MoveJob &j = ride.job;
    fill(j.a5, l_small0, n_small, Bs); j.n5 = (unsigned)((size_t)n_small * njt * nch * 4 * (H / I8_PD));
    if (l_big >= 0) { fill(j.a6, l_big, 1, Bb); j.n6 = (unsigned)((size_t)njt * nch * 4 * (H / I8_PD)); } else { memset(&j.a6, 0, sizeof j.a6); j.n6 = 0; }
    j.nblocks = (unsigned)ctx->cfg.pt_ride; j.depth = 1; j.nt = 1; j.first = 0; j.count = 0;
#ifdef SFG_AB
    j.depth = ctx->cfg.i8_mover_depth_ride; j.nt = ctx->cfg.i8_mover_nt_ride;
#endif
    ride.next = 0; ride.per = (ride.total() + (unsigned)launches - 1) / (unsigned)launches;
    ride.item_bytes5 = 5.0 * 2 * 32768; ride.item_bytes6 = 6.0 * 2 * 32768;         // a unit = one digit plane of an item: 256 rows of 128 bytes in, 128 pieces of 256 bytes out
    ride.on = true;
    return 0;
}
int i8_ride_finish(sfg_ctx *ctx, PtRide &ride) {
    if (!ride.on || ride.next >= ride.total()) return 0;
    MoveJob j = ride.job; j.first = ride.next; j.count = ride.total() - ride.next; ride.next = ride.total();
    j.nblocks = std::min(2048u, (j.count + 7u) / 8u * 8u);                       // alone on the chip the mover wants many workgroups (profiles/r06_mover_ubench.txt)
    PhaseTimer t(ctx, "mac_i8_pack_pt");
    launch_move_alone(ctx->stream, j);
    SFG_HIP(ctx, hipGetLastError());
    t.stop(1);
    return 0;
}
template <int ND>
static int launch_mac_i8_nd(sfg_ctx *ctx, const double *rotf, size_t rotf_k_stride, size_t rotf_r_stride, int plane0, const u64 *pt, u64 *out, int K, int R, int r0, int Ncols,
                            int l0, int nl, int accumulate, const MacStrides &st) {
    const int N = SFG_N, H = N / 2;
    constexpr bool BIG = ND == 6;
    if (!st.pt_half || (!BIG && !st.pt_packed)) SFG_FAIL(ctx, "sfg_mac (i8): needs half-row plaintext rows (packed-limb words or digit planes for the small moduli)");
    if (BIG && nl != 1) SFG_FAIL(ctx, "sfg_mac (i8): one 46-bit modulus per launch");
    if (Ncols > 96) SFG_FAIL(ctx, "sfg_mac (i8): more than 96 columns per launch");
    if ((long long)K * ND >= 131072) SFG_FAIL(ctx, "sfg_mac (i8): K too large for the int32 digit sums (ND K 2^14 must stay below 2^31)");
    // digit range: ND signed base-256 digits hold -0x80..80 <= v <= 0x7F..7F - the canonical plaintext word (< q) and the centred rot word (|v| <= q / 2); the Horner
    // recombination (i8_horner) is exact below 2^47 (two x 16 steps above 2^46 - 2^24), and the five-digit kernels use its one-step form only (128 q < 2^53 with room to spare)
    for (int t = l0; t < l0 + nl; t++) {
        if (ctx->q[t] > (BIG ? SFG_I8_BIG_QMAX : (1ULL << 38))) SFG_FAIL(ctx, "sfg_mac (i8): modulus %d = %llu does not fit %d signed base-256 digits / the exact fp64 recombination", t, (unsigned long long)ctx->q[t], ND);
    }
    const int8_t *B_given = BIG ? st.B_big : st.B_small;            // B_mode 0: streamed transposition, the plaintext tiles are in place, k' = g * kb + baby;
    const bool B_stream = B_given && st.B_mode == 0;                // 1: in place in the pass's own layout (the riding mover made them); 2: the pass below runs into this buffer
    const int8_t *B_pre = B_given && st.B_mode != 2 ? B_given : nullptr;
    if (B_given && !B_stream && !st.pt_digits) SFG_FAIL(ctx, "sfg_mac (i8): internal: given tile buffers take digit-plane panels");
    if (B_stream && (K % SFG_D || !st.kb || Ncols != SFG_D)) SFG_FAIL(ctx, "sfg_mac (i8): internal: prepacked tiles need whole block rows and 91 columns");
    const int8_t *A_pre = BIG ? st.A_big : st.A_small;             // the transposed rot tiles of exactly this launch, made by launch_i8_pack_rot_to (I8RotPre)
    if (A_pre && (B_stream || r0 != 0 || R > 32)) SFG_FAIL(ctx, "sfg_mac (i8): internal: given rot tiles cover one block of <= 32 rows, with the plaintext panel");
    const int K_rot = K;                                            // rows of the rot operand
    if (B_stream) K = K / SFG_D * st.kb;
    I8Args a; memset(&a, 0, sizeof a); a.kb = B_stream ? st.kb : 0;
    a.rotf = rotf; a.pt = pt; a.out = out; a.rotf_k_stride = rotf_k_stride; a.rotf_r_stride = rotf_r_stride;
    a.pt_k_stride = st.pt_k; a.pt_n_stride = st.pt_n; a.out_n_stride = st.out_n; a.out_r_stride = st.out_r;
    if (st.pt_layout && !st.pt_digits) SFG_FAIL(ctx, "sfg_mac (i8): internal: compact panel rows hold digit planes");
    if (st.pt_layout == 2 && B_stream) SFG_FAIL(ctx, "sfg_mac (i8): internal: a K-major panel with streamed tiles");
    i8_panel_rows(ctx, st.pt_layout, K, st.pt_L, l0, BIG, a);
    a.K = K; a.R = R; a.Ncols = Ncols; a.accumulate = accumulate; a.r0 = r0; a.l0 = l0; a.nl = nl; a.plane0 = plane0;
    a.nch = (K + 63) / 64; a.njt = (Ncols + 15) / 16; a.pt_digits = st.pt_digits ? 1 : 0;
    const size_t nA = (size_t)nl * N * a.nch * 2 * ND * 1024, nB = (size_t)nl * H * a.njt * a.nch * ND * 1024, nT = (size_t)nl * H * 2 * a.njt * 2 * 256;
    // the transposed rot operand is kept while its source (pointer, generation, shape) is unchanged: a group's rotation cache serves every block column.
    // Two copies per kind of modulus (the pipelined product alternates between two rot buffers).
    const u64 sig[8] = {ctx->i8_gen, (u64)K | (u64)a.kb << 32, (u64)R, (u64)r0, (u64)l0 << 8 | (u64)nl, (u64)plane0, (u64)rotf_k_stride, (u64)rotf_r_stride};
    sfg_ctx::I8Slot *slots = ctx->i8_slot[BIG ? 1 : 0];
    int slot = -1;
    if (!A_pre) for (int i = 0; i < sfg_ctx::I8_SLOTS; i++) if (slots[i].src == (const void *)rotf && !memcmp(slots[i].sig, sig, sizeof sig)) slot = i;
    const bool repack = slot < 0 && !A_pre;
    char nm[24];
    if (repack) {
        // victim: a copy of a stale generation (a product's earlier group), else an unused slot if the HBM takes another copy, else the least recently used one
        int stale = -1, empty = -1, lru = 0;
        for (int i = 0; i < sfg_ctx::I8_SLOTS; i++) {
            if (!slots[i].src) { if (empty < 0) empty = i; continue; }
            if (slots[i].sig[0] != ctx->i8_gen && (stale < 0 || slots[i].last_use < slots[stale].last_use)) stale = i;
            if (slots[i].last_use < slots[lru].last_use || !slots[lru].src) lru = i;
        }
        slot = stale;
        if (slot < 0 && empty >= 0) {
            int live = 0; for (int i = 0; i < sfg_ctx::I8_SLOTS; i++) live += slots[i].src != nullptr;
            size_t fr = 0, tot = 0;
            if (live < 2 || (hipMemGetInfo(&fr, &tot) == hipSuccess && fr >= nA + (48ULL << 30))) slot = empty;
        }
        if (slot < 0) slot = lru;
    }
    if (A_pre) a.A = const_cast<int8_t *>(A_pre);
    else {
        snprintf(nm, sizeof nm, BIG ? "mi8.Ab%d" : "mi8.A%d", slot);
        SFG_TRY(sfg_scratch(ctx, nm, nA, (void **)&a.A));
        if (repack) { slots[slot].src = (const void *)rotf; memcpy(slots[slot].sig, sig, sizeof sig); }
        slots[slot].last_use = ++ctx->i8_clock;
    }
    if (B_given) a.B = const_cast<int8_t *>(B_given);
    else SFG_TRY(sfg_scratch(ctx, !ctx->cfg.stage_pack ? "mi8.B" : BIG ? "mi8.Bb" : "mi8.Bs", nB, (void **)&a.B));        // (with the streamed transposition on: the buffers of the streamed tiles, a launch uses them one way or the other)
    SFG_TRY(sfg_scratch(ctx, "mi8.T", nT * 8, (void **)&a.T));
    // (a regrown B / T buffer belongs to this launch alone; the A copies have their own buffers)
    if (!A_pre && !repack && ctx->pool[nm].second < nA) SFG_FAIL(ctx, "sfg_mac (i8): internal: kept rot copy smaller than its operand");
    const double tile = 1024.0;
    if (repack) { PhaseTimer t(ctx, "mac_i8_pack_rot");
      hipLaunchKernelGGL(k_i8_pack_rot<ND>, dim3((unsigned)((size_t)nl * a.nch * (N / I8_PC) * 4)), dim3(256), I8_PC * i8_rot_img_stride<ND>(), ctx->stream, a);
      SFG_HIP(ctx, hipGetLastError()); t.stop(1, (double)nl * N * ((double)K_rot * std::min(32, R - r0) * (BIG ? 16.0 : 8.0) + (double)a.nch * 2 * ND * tile)); }
    if (!B_pre) { PhaseTimer t(ctx, "mac_i8_pack_pt");
      const unsigned items = (unsigned)((size_t)nl * a.njt * a.nch * 4 * (H / I8_PD));
#ifdef SFG_AB
      if (a.pt_digits && ctx->cfg.i8_mover > 0 && a.pt_n_stride * 8 < (1ULL << 31)) {      // the mover form of the pass (i8_move.hpp): fewer, longer-lived workgroups with the next units' loads in flight
          MoveJob j; memset(&j.a5, 0, sizeof j.a5); memset(&j.a6, 0, sizeof j.a6);
          if (ND == 5) { j.a5 = a; j.n5 = items; } else { j.a6 = a; j.n6 = items; }
          j.first = 0; j.count = items; j.nblocks = std::min((unsigned)ctx->cfg.i8_mover, (items + 7u) / 8u * 8u); j.depth = ctx->cfg.i8_mover_depth; j.nt = 0;
          launch_move_alone(ctx->stream, j);
      } else
#endif
      if (a.pt_digits) hipLaunchKernelGGL(k_i8_pack_pt_digits<ND>, dim3(items), dim3(256), 0, ctx->stream, a);
      else hipLaunchKernelGGL(k_i8_pack_pt<ND>, dim3((unsigned)((size_t)nl * a.njt * a.nch * 4 * (H / I8_PP))), dim3(256), 0, ctx->stream, a);
      SFG_HIP(ctx, hipGetLastError()); t.stop(1, (double)nl * H * ((double)K * Ncols * (a.pt_digits ? (double)ND : 8.0) + (double)a.njt * a.nch * ND * tile)); }
    { PhaseTimer t(ctx, BIG ? "mac_big" : "mac_small");            // the MAC proper: both operand streams read once, tile-ordered results written
      if (a.njt == 6 && ctx->cfg.mac_i8_ring) {
          if (BIG && ctx->cfg.mac_i8_waves == 12) hipLaunchKernelGGL((k_mac_i8_ring<6, 2, 0, 2>), dim3((unsigned)(nl * H)), dim3(768), 2 * 10 * 6 * 1024, ctx->stream, a, ctx->modc);
          else if (BIG) hipLaunchKernelGGL((k_mac_i8_ring<6, 2>), dim3((unsigned)(nl * H)), dim3(384), 2 * 10 * 6 * 1024, ctx->stream, a, ctx->modc);
          else if (ctx->cfg.mac_i8_waves == 12 && !ctx->cfg.mac_i8_diag) hipLaunchKernelGGL((k_mac_i8_ring<5, 3, 0, 2>), dim3((unsigned)(nl * H)), dim3(768), 3 * 10 * 5 * 1024, ctx->stream, a, ctx->modc);
#ifdef SFG_AB
          else if (ctx->cfg.mac_i8_diag == 1) hipLaunchKernelGGL((k_mac_i8_ring<5, 3, 1>), dim3((unsigned)(nl * H)), dim3(384), 3 * 10 * 5 * 1024, ctx->stream, a, ctx->modc);
          else if (ctx->cfg.mac_i8_diag == 2) hipLaunchKernelGGL((k_mac_i8_ring<5, 3, 2>), dim3((unsigned)(nl * H)), dim3(384), 3 * 10 * 5 * 1024, ctx->stream, a, ctx->modc);
#endif
          else hipLaunchKernelGGL((k_mac_i8_ring<5, 3>), dim3((unsigned)(nl * H)), dim3(384), 3 * 10 * 5 * 1024, ctx->stream, a, ctx->modc);
      }
#ifdef SFG_AB
      else if (!BIG && a.njt == 6 && !ctx->cfg.mac_i8_nolds) hipLaunchKernelGGL(k_mac_i8_lds, dim3((unsigned)(nl * H)), dim3(384), 0, ctx->stream, a, ctx->modc);
#endif
      else if (ctx->cfg.mac_i8_wg1) hipLaunchKernelGGL(k_mac_i8<ND>, dim3((unsigned)(nl * H * a.njt)), dim3(64), 0, ctx->stream, a, ctx->modc);
      else hipLaunchKernelGGL(k_mac_i8<ND>, dim3((unsigned)(nl * H)), dim3(64 * a.njt), 0, ctx->stream, a, ctx->modc);
      SFG_HIP(ctx, hipGetLastError());
      t.stop(1, (double)nl * ((double)N * a.nch * 2 * ND * tile + (double)H * a.njt * a.nch * ND * tile + (double)H * 2 * a.njt * 2 * 256 * 8.0)); }
    { PhaseTimer t(ctx, "mac_i8_untile");
      hipLaunchKernelGGL(k_i8_untile, dim3((unsigned)((size_t)nl * (H / 16) * 2 * a.njt * 2)), dim3(256), 0, ctx->stream, a, ctx->modc);
      SFG_HIP(ctx, hipGetLastError());
      t.stop(1, (double)nl * N * (double)Ncols * std::min(32, R - r0) * 8.0 * (accumulate ? 3.0 : 2.0)); }
    return 0;
}
// the transposed rot tiles of one MAC group into a caller's buffer (mac_i8_rot_tile_bytes(K, nl, ND) bytes): rows [0, R) of every k-slice, R <= 32
size_t mac_i8_rot_tile_bytes(int K, int nl, int ND) { return (size_t)nl * SFG_N * (((size_t)K + 63) / 64) * 2 * ND * 1024; }
int launch_i8_pack_rot_to(sfg_ctx *ctx, const double *rotf, size_t rotf_k_stride, size_t rotf_r_stride, int plane0, int K, int R, int l0, int nl, bool big, int8_t *A_out) {
    if (R > 32 || (big && nl != 1)) SFG_FAIL(ctx, "i8 rot tiles: internal: at most 32 rows, one 46-bit modulus per buffer");
    I8Args a; memset(&a, 0, sizeof a);
    a.rotf = rotf; a.rotf_k_stride = rotf_k_stride; a.rotf_r_stride = rotf_r_stride; a.K = K; a.R = R; a.r0 = 0; a.l0 = l0; a.nl = nl; a.plane0 = plane0;
    a.nch = (K + 63) / 64; a.kb = 0; a.A = A_out;
    const int N = SFG_N;
    PhaseTimer t(ctx, "mac_i8_pack_rot");
    if (big) hipLaunchKernelGGL(k_i8_pack_rot<6>, dim3((unsigned)((size_t)nl * a.nch * (N / I8_PC) * 4)), dim3(256), I8_PC * i8_rot_img_stride<6>(), ctx->stream, a);
    else hipLaunchKernelGGL(k_i8_pack_rot<5>, dim3((unsigned)((size_t)nl * a.nch * (N / I8_PC) * 4)), dim3(256), I8_PC * i8_rot_img_stride<5>(), ctx->stream, a);
    SFG_HIP(ctx, hipGetLastError());
    t.stop(1);
    return 0;
}
int launch_mac_i8_small(sfg_ctx *ctx, const double *rotf, size_t rotf_k_stride, size_t rotf_r_stride, int plane0, const u64 *pt, u64 *out, int K, int R, int r0, int Ncols,
                        int l0, int nl, int accumulate, const MacStrides &st) {
    return launch_mac_i8_nd<5>(ctx, rotf, rotf_k_stride, rotf_r_stride, plane0, pt, out, K, R, r0, Ncols, l0, nl, accumulate, st);
}
int launch_mac_i8_big(sfg_ctx *ctx, const double *rotf, size_t rotf_k_stride, size_t rotf_r_stride, int plane0, const u64 *pt, u64 *out, int K, int R, int r0, int Ncols,
                      int l0, int accumulate, const MacStrides &st) {
    return launch_mac_i8_nd<6>(ctx, rotf, rotf_k_stride, rotf_r_stride, plane0, pt, out, K, R, r0, Ncols, l0, 1, accumulate, st);
}

// ---- test hook: the product's default MAC on caller-given operands (tests/test_gpu_mac.py; refused without SFG_ENABLE_TEST_HOOKS=1)
// half rows of canonical words -> the digit-plane rows the plaintext NTT writes (k_ntt_half3<., true>): plane d of row (k, n, l) holds byte c = digit d of word c
__global__ void __launch_bounds__(256) k_i8_words_to_planes(const u64 *in, u64 *out, int L, unsigned big_mask) {
    const int H = SFG_N / 2;
    const size_t row = blockIdx.x; const int l = (int)(row % (size_t)L);
    const u64 *src = in + row * H; int8_t *dst = reinterpret_cast<int8_t *>(out + row * H);
    const bool big = (big_mask >> l) & 1u;
    for (int c = threadIdx.x; c < H; c += 256) {
        const long long v = (long long)src[c];
        if (big) { int8_t d[6]; i8_digits<6>(v, d);
#pragma unroll
            for (int i = 0; i < 6; i++) dst[(size_t)i * H + c] = d[i]; }
        else { int8_t d[5]; i8_digits<5>(v, d);
#pragma unroll
            for (int i = 0; i < 5; i++) dst[(size_t)i * H + c] = d[i]; }
    }
}
extern "C" int sfg_mac_i8_dev(sfg_ctx *ctx, const uint64_t *rot, const uint64_t *pt_half, uint64_t *out, int K, int R, int Ncols, int L, int accumulate, int pt_form) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->test_hooks) SFG_FAIL(ctx, "sfg_mac_i8_dev is a test hook: the context was not created under the test switch");
    if (L < 1 || L > ctx->nq) SFG_FAIL(ctx, "sfg_mac_i8: L out of range");
    if (K < 1 || R < 1 || Ncols < 1) SFG_FAIL(ctx, "sfg_mac_i8: K, R and Ncols must be positive (got %d, %d, %d)", K, R, Ncols);
    if (Ncols > 96 || (long long)K * 6 >= 131072) SFG_FAIL(ctx, "sfg_mac_i8: at most 96 columns and K < 21846 per launch");
    if (pt_form != 0 && pt_form != 1) SFG_FAIL(ctx, "sfg_mac_i8: pt_form is 0 (digit planes) or 1 (panel words)");
    const int N = SFG_N, H = N / 2;
    std::vector<int> plane_of, is_big; const int nplanes = mac_dma_planes(ctx, L, plane_of, is_big);
    if (nplanes < 0) return 1;
    unsigned big_mask = 0, small_mask = 0; for (int l = 0; l < L; l++) (is_big[l] ? big_mask : small_mask) |= 1u << l;
    if (big_mask && !ctx->cfg.mac_i8_big) SFG_FAIL(ctx, "sfg_mac_i8: a modulus above 2^36 is on the fp64 kernel in this context (it exceeds six signed digits: q > 0x7F7F7F7F7F80; or the A/B build's switch)");
    const size_t rrows = (size_t)K * R, prows = (size_t)K * Ncols * L;
    double *rotf = nullptr; u64 *ptp = nullptr;
    int rc = 0;
    if (hipMalloc(&rotf, rrows * (size_t)nplanes * N * 8) != hipSuccess || hipMalloc(&ptp, prows * H * 8) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(rotf); SFG_FAIL(ctx, "sfg_mac_i8: out of device memory"); }
    rc = launch_rot_to_f64(ctx, (const u64 *)rot, rrows, L, L, rotf);                   // centred fp64 planes, {lo 23 bits, hi} pairs for a big modulus: the product's rot operand
    if (!rc) {
        if (pt_form == 0) { hipLaunchKernelGGL(k_i8_words_to_planes, dim3((unsigned)prows), dim3(256), 0, ctx->stream, (const u64 *)pt_half, ptp, L, big_mask); if (hipGetLastError() != hipSuccess) { rc = 1; ctx->err = "sfg_mac_i8: plane kernel launch failed"; } }
        else rc = launch_pack_pt(ctx, (const u64 *)pt_half, ptp, prows, (size_t)H, L, small_mask);       // packed-limb words for the small moduli, plain words for a big one (the DiagCache product's panel)
    }
    if (!rc) {
        MacStrides st;
        st.rot_k = (size_t)R * L * N; st.rot_r = (size_t)L * N;
        st.pt_k = (size_t)Ncols * L * H; st.pt_n = (size_t)L * H;
        st.out_n = (size_t)R * L * N; st.out_r = (size_t)L * N;
        st.pt_half = true; st.pt_packed = true; st.i8 = st.i8_big = true; st.pt_digits = st.pt_digits_big = pt_form == 0;
        PhaseTimer t(ctx, "mac");
        rc = launch_mac_bc(ctx, rotf, (size_t)R, ptp, (u64 *)out, K, R, Ncols, L, accumulate, st, nullptr);
        t.stop(1);
    }
    (void)hipStreamSynchronize(ctx->stream); (void)hipFree(rotf); (void)hipFree(ptp);
    return rc;
}

#ifdef SFG_AB
// ---- round 6: the mover (i8_move.hpp) against the pass, alone and in front of the plaintext NTT's workgroups.  Test hooks, not part of the C-ABI header.
__global__ void __launch_bounds__(256) k_ub_fill(u64 *p, size_t n, u64 seed) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        u64 z = (i + seed) * 0x9E3779B97F4A7C15ULL; z ^= z >> 29; z *= 0xBF58476D1CE4E5B9ULL; z ^= z >> 32; p[i] = z;
    }
}
__global__ void __launch_bounds__(256) k_ub_diff(const u64 *a, const u64 *b, size_t n, unsigned long long *cnt) {
    unsigned long long c = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) c += a[i] != b[i];
    if (c) atomicAdd(cnt, c);
}
static int move_job_for(sfg_ctx *ctx, const u64 *panel, int G, int L, int8_t *Bs, int8_t *Bb, MoveJob &pj) {
    const int H = SFG_N / 2, d = SFG_D;
    std::vector<int> plane_of, is_big; if (mac_dma_planes(ctx, L, plane_of, is_big) < 0) return 1;
    int l_big = -1, l_small0 = -1, n_small = 0;
    for (int l = 0; l < L; l++) { if (is_big[l]) l_big = l; else { if (l_small0 < 0) l_small0 = l; n_small++; } }
    if (l_big < 0 || n_small != L - 1) SFG_FAIL(ctx, "mover: expects one 46-bit modulus and 35-bit ones");
    const size_t plw = (size_t)L * H;
    const int K = G * d, nch = (K + 63) / 64;
    auto fill = [&](I8Args &a, int l0, int nl, int8_t *B) {
        memset(&a, 0, sizeof a);
        a.pt = panel; a.pt_k_stride = plw; a.pt_n_stride = (size_t)G * d * plw; a.pt_l_stride = H; a.pt_l0_off = (size_t)l0 * H; a.pt_d_stride = H; a.pt_cb_stride = I8_PD; a.K = K; a.Ncols = d; a.l0 = l0; a.nl = nl; a.nch = nch; a.njt = 6; a.pt_digits = 1; a.B = B;
    };
    fill(pj.a5, l_small0, n_small, Bs); fill(pj.a6, l_big, 1, Bb);
    pj.n5 = (unsigned)(n_small * 6 * nch * 4 * (H / I8_PD)); pj.n6 = (unsigned)(6 * nch * 4 * (H / I8_PD));
    if (pj.a5.pt_n_stride * 8 >= (1ULL << 32) / 2) SFG_FAIL(ctx, "mover: panel column stride does not fit the 32-bit lane offset");
    return 0;
}
// mode 0: NTTs then the pass; 2: NTTs alone; 3: the pass alone; 4: the mover alone (nblocks workgroups); 5: mover workgroups in front of every NTT launch;
// 6: check - random panel bytes through the pass and through the mover (alone), *ms_out = number of differing tile words (0 = identical);
// 7: the same with the mover riding in NTT launches (the NTT writes another panel).  G block rows of 8281 plaintexts, cfg.enc_batch plaintexts per NTT launch.
extern "C" int ubench_ntt_move(sfg_ctx *ctx, int mode, int G, int nblocks, int depth, int nt, int reps, double *ms_out) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (!ctx->test_hooks) SFG_FAIL(ctx, "ubench_ntt_move is a test hook");
    if (G < 1 || G > 24 || nblocks < 8 || nblocks % 8 || depth < 1 || depth > 3) SFG_FAIL(ctx, "ubench_ntt_move: bad arguments");
    const int N = SFG_N, H = N / 2, L = 5, d = SFG_D;
    const size_t plw = (size_t)L * H, nplain = (size_t)d * d, batch = (size_t)ctx->cfg.enc_batch;
    const size_t total = (size_t)G * nplain; const int launches = (int)((total + batch - 1) / batch);
    const bool check = mode == 6 || mode == 7;
    double *pc; u64 *panel, *panel2 = nullptr; int8_t *Bs, *Bb, *Bs2 = nullptr, *Bb2 = nullptr;
    const int K = G * d;
    const size_t nBs = mac_i8_tile_bytes(K, 4, 5), nBb = mac_i8_tile_bytes(K, 1, 6);
    SFG_TRY(sfg_scratch(ctx, "ub.pc", batch * H * 8, (void **)&pc));
    SFG_TRY(sfg_scratch(ctx, "ub.pt", total * plw * 8 + (1 << 20), (void **)&panel));
    SFG_TRY(sfg_scratch(ctx, "ub.Bs", nBs, (void **)&Bs));
    SFG_TRY(sfg_scratch(ctx, "ub.Bb", nBb, (void **)&Bb));
    if (check || mode == 5) SFG_TRY(sfg_scratch(ctx, "ub.pt2", total * plw * 8 + (1 << 20), (void **)&panel2));
    if (check) { SFG_TRY(sfg_scratch(ctx, "ub.Bs2", nBs, (void **)&Bs2)); SFG_TRY(sfg_scratch(ctx, "ub.Bb2", nBb, (void **)&Bb2)); }
    SFG_HIP(ctx, hipMemsetAsync(pc, 0, batch * H * 8, ctx->stream));
    MoveJob pj; SFG_TRY(move_job_for(ctx, panel, G, L, Bs, Bb, pj));
    pj.nblocks = (unsigned)nblocks; pj.depth = depth; pj.nt = nt;
    if (const char *e = getenv("SFG_UB_MOVER_FAKE")) pj.a5.fake = pj.a6.fake = atoi(e);          // timing experiments with INVALID results (i8_move.hpp)
    PanelMap pm; pm.G = 0; pm.g = 0; pm.shift0 = 0; pm.packed_mask = mac_dma_packed_mask(ctx, L) | 0x80000000u | 0x40000000u;
    if (pj.a5.fake & 16) { pm.packed_mask |= PT_COMPACT | PT_KMAJOR; pm.K = K; }                  // the NTTs write the K-major panel pattern (the launch's plaintexts: column p / K, row p % K)
    auto pack_alone = [&](const MoveJob &j) {
        hipLaunchKernelGGL(k_i8_pack_pt_digits<5>, dim3(j.n5), dim3(256), 0, ctx->stream, j.a5);
        hipLaunchKernelGGL(k_i8_pack_pt_digits<6>, dim3(j.n6), dim3(256), 0, ctx->stream, j.a6);
    };
    // NTT launches (into ntt_out) with the job's items spread evenly over them
    auto ntts = [&](u64 *ntt_out, const MoveJob *mv) -> int {
        unsigned next = 0; const unsigned all = mv ? mv->n5 + mv->n6 : 0u, per = mv ? (all + launches - 1) / launches : 0u;
        for (int i = 0; i < launches; i++) {
            const size_t lo = (size_t)i * batch, nb = std::min(batch, total - lo);
            MoveJob j; if (mv) { j = *mv; j.first = next; j.count = std::min(per, all - next); next += j.count; }
            SFG_TRY(launch_ntt_plain_half(ctx, pc, ntt_out + lo * plw, nb, L, pm, nullptr, mv ? &j : nullptr));
        }
        if (mv && next != all) SFG_FAIL(ctx, "ubench_ntt_move: %u items left over", all - next);
        return 0;
    };
    if (check) {
        hipLaunchKernelGGL(k_ub_fill, dim3(4096), dim3(256), 0, ctx->stream, panel, total * plw, (u64)G * 977u);
        SFG_HIP(ctx, hipMemsetAsync(Bs, 0x5A, nBs, ctx->stream)); SFG_HIP(ctx, hipMemsetAsync(Bb, 0x5A, nBb, ctx->stream));
        SFG_HIP(ctx, hipMemsetAsync(Bs2, 0xA5, nBs, ctx->stream)); SFG_HIP(ctx, hipMemsetAsync(Bb2, 0xA5, nBb, ctx->stream));
        pack_alone(pj);
        MoveJob j2 = pj; j2.a5.B = Bs2; j2.a6.B = Bb2; j2.first = 0; j2.count = pj.n5 + pj.n6;
        if (mode == 6) launch_move_alone(ctx->stream, j2); else SFG_TRY(ntts(panel2, &j2));
        unsigned long long *cnt; SFG_TRY(sfg_scratch(ctx, "ub.cnt", 8, (void **)&cnt));
        SFG_HIP(ctx, hipMemsetAsync(cnt, 0, 8, ctx->stream));
        hipLaunchKernelGGL(k_ub_diff, dim3(4096), dim3(256), 0, ctx->stream, (const u64 *)Bs, (const u64 *)Bs2, nBs / 8, cnt);
        hipLaunchKernelGGL(k_ub_diff, dim3(4096), dim3(256), 0, ctx->stream, (const u64 *)Bb, (const u64 *)Bb2, nBb / 8, cnt);
        unsigned long long h = 0;
        SFG_HIP(ctx, hipMemcpyAsync(&h, cnt, 8, hipMemcpyDeviceToHost, ctx->stream)); SFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        *ms_out = (double)h;
        return 0;
    }
    auto run = [&]() -> int {
        if (mode == 0 || mode == 2) SFG_TRY(ntts(panel, nullptr));
        if (mode == 0 || mode == 3) pack_alone(pj);
        if (mode == 4) { MoveJob j = pj; j.first = 0; j.count = pj.n5 + pj.n6; launch_move_alone(ctx->stream, j); }
        if (mode == 5) SFG_TRY(ntts(panel2, &pj));
        SFG_HIP(ctx, hipGetLastError());
        return 0;
    };
    hipEvent_t e0, e1; SFG_HIP(ctx, hipEventCreate(&e0)); SFG_HIP(ctx, hipEventCreate(&e1));
    SFG_TRY(run());
    SFG_HIP(ctx, hipEventRecord(e0, ctx->stream));
    for (int r = 0; r < reps; r++) SFG_TRY(run());
    SFG_HIP(ctx, hipEventRecord(e1, ctx->stream));
    SFG_HIP(ctx, hipEventSynchronize(e1));
    float ms = 0; SFG_HIP(ctx, hipEventElapsedTime(&ms, e0, e1));
    *ms_out = (double)ms / reps;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return 0;
}
#endif
