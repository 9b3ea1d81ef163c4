// ntt_core.hpp — register/LDS phases of the 16384-point negacyclic forward NTT shared by ntt.hip and the fused key-switch
// tail in rotate.hip.  See ntt.hip for the index split j = a*512 + b*16 + c and the LDS image.
#pragma once
#include "common.hpp"

constexpr int LDS_ROW = 528;                 // 512 + 16 doubles: keeps (a, a+1) rows 32 banks apart
constexpr int LDS_DOUBLES = 32 * LDS_ROW;    // 135,168 B

// twiddles are stored as w only; the quotient of a butterfly product is estimated from the rounded product and 1/q (mulmod_lazy_q)
template <int LEN, int H, class TW>
__device__ __forceinline__ void ct_stage(double (&v)[LEN], double q, double qinv, TW tw) {
#pragma unroll
    for (int g = 0; g < LEN / (2 * H); g++) {
        const double w = tw(g);
#pragma unroll
        for (int x = 0; x < H; x++) {
            const int i0 = g * 2 * H + x, i1 = i0 + H;
            double r = mulmod_lazy_q(v[i1], w, q, qinv);
            double U = v[i0];
            v[i0] = U + r; v[i1] = U - r;
        }
    }
}
template <int LEN, int H, class TW>
__device__ __forceinline__ void gs_stage(double (&v)[LEN], double q, double qinv, TW tw) {
#pragma unroll
    for (int g = 0; g < LEN / (2 * H); g++) {
        const double w = tw(g);
#pragma unroll
        for (int x = 0; x < H; x++) {
            const int i0 = g * 2 * H + x, i1 = i0 + H;
            double U = v[i0], V = v[i1];
            v[i0] = U + V;
            v[i1] = mulmod_lazy_q(U - V, w, q, qinv);
        }
    }
}

// Forward phases A, B, C on the 32 values v[a] = x[a*512 + tid] of one 512-thread workgroup.  On return (after the final
// barrier) the lazy result for output index j = a*512 + b*16 + c sits at lds[a*LDS_ROW + c*33 + b].
__device__ __forceinline__ void ntt_fwd_phases(double (&v)[32], double *lds, const double *tw, const double2 *pack, double q, double qinv, int tid) {
    ct_stage<32, 16>(v, q, qinv, [&](int g) { return tw[1 + g]; });
    ct_stage<32, 8>(v, q, qinv, [&](int g) { return tw[2 + g]; });
    ct_stage<32, 4>(v, q, qinv, [&](int g) { return tw[4 + g]; });
    ct_stage<32, 2>(v, q, qinv, [&](int g) { return tw[8 + g]; });
    ct_stage<32, 1>(v, q, qinv, [&](int g) { return tw[16 + g]; });
#pragma unroll
    for (int a = 0; a < 32; a++) lds[a * LDS_ROW + tid] = v[a];
    __syncthreads();
    {   // phase B: thread (a, c), local b
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
    // phase C: two (a, b) groups per thread, local c
#pragma unroll
    for (int h = 0; h < 2; h++) {
        const int p = tid + 512 * h, a = p >> 5, b = p & 31;
        double w[16];
#pragma unroll
        for (int c = 0; c < 16; c++) w[c] = lds[a * LDS_ROW + c * 33 + b];
        // the 15 twiddles of group ab = p (1 + 2 + 4 + 8) come from 8 fully coalesced 16-byte loads:
        // pack[(p / 64)][i][lane] holds entries {2i, 2i+1} of the list [T8, T4_0, T4_1, T2_0..3, T1_0..7, pad]
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
        const int p = tid + 512 * h, a = p >> 5, b = p & 31;
#pragma unroll
        for (int c = 0; c < 16; c++) lds[a * LDS_ROW + c * 33 + b] = v[h * 16 + c];
    }
    __syncthreads();
}
