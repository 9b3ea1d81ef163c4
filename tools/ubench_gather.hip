// Piece-gather / piece-scatter bandwidth probe for the transposition kernels (k_i8_pack_pt_digits reads 128-byte pieces of 256 rows 320 KiB apart and writes
// 256-byte pieces ~690 KiB apart): what run length does HBM want at these strides?
//   ./ubench_gather <r|w> <piece bytes P> <row stride bytes S> <span bytes per row> <total GB> <bytes per lane 4|16> [order 0|1]
// A workgroup (256 threads) moves 256 rows x P bytes at column offset piece * P.  order 0: consecutive workgroups take consecutive pieces of the same rows
// (the pack kernel's order), order 1: consecutive workgroups take consecutive row groups of the same piece.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
template <int V, bool WR>
__global__ void __launch_bounds__(256) k_piece(unsigned char *buf, size_t S, int P, int npiece, int ngroups, int order, unsigned *sink, size_t S2, int nkq) {
    const int b = blockIdx.x;
    const int piece = order ? b / ngroups : b % npiece, grp = order ? b % ngroups : b / npiece;
    const int lpr = P / V, rpp = 256 / lpr;                 // lanes per row, rows per pass
    const int tid = threadIdx.x, lr = tid / lpr, lc = tid % lpr;
    unsigned acc = 0;
    // S2 = 0: 256 consecutive rows S apart.  S2 > 0 (the pack kernel's shape): group = (kq, jt), rows = 16 columns S2 apart x 16 consecutive k S apart
    unsigned char *base = buf + (S2 ? (size_t)(grp / nkq) * 16 * S2 + (size_t)(grp % nkq) * 16 * S : (size_t)grp * 256 * S) + (size_t)piece * P + (size_t)lc * V;
#pragma unroll 8
    for (int r = lr; r < 256; r += rpp) {
        unsigned char *p = base + (S2 ? (size_t)(r >> 4) * S2 + (size_t)(r & 15) * S : (size_t)r * S);
        if (V == 4) { if (WR) *reinterpret_cast<unsigned *>(p) = tid + r; else acc ^= *reinterpret_cast<const unsigned *>(p); }
        else { if (WR) *reinterpret_cast<uint4 *>(p) = make_uint4(tid, r, b, 0); else { const uint4 w = *reinterpret_cast<const uint4 *>(p); acc ^= w.x ^ w.y ^ w.z ^ w.w; } }
    }
    if (!WR && acc == 0x12345679u) sink[0] = acc;
}
int main(int argc, char **argv) {
    if (argc < 7) { printf("usage: r|w P S span GB V [order]\n"); return 1; }
    const bool wr = argv[1][0] == 'w';
    const int P = atoi(argv[2]); const size_t S = (size_t)atoll(argv[3]); const int span = atoi(argv[4]); const double gb = atof(argv[5]); const int V = atoi(argv[6]);
    const int order = argc > 7 ? atoi(argv[7]) : 0;
    const int nk = argc > 8 ? atoi(argv[8]) : 0;          // > 0: the pack kernel's shape, 96 columns x nk rows (nk a multiple of 16), columns nk * S apart
    if (P % V || 256 % (P / V) || P / V > 256 || span % P || (size_t)span > S) { printf("bad shape\n"); return 1; }
    const size_t rows = nk ? (size_t)96 * nk : (size_t)(gb * 1e9 / S) / 256 * 256;
    const int ngroups = (int)(rows / 256), npiece = span / P;
    const size_t S2 = nk ? (size_t)nk * S : 0; const int nkq = nk ? nk / 16 : 1;
    unsigned char *buf; unsigned *sink;
    CK(hipMalloc(&buf, rows * S)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(buf, 1, rows * S));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const dim3 grid((unsigned)((size_t)ngroups * npiece));
    float best = 1e30f;
    for (int it = 0; it < 4; it++) {
        CK(hipEventRecord(e0));
        if (V == 4) { if (wr) hipLaunchKernelGGL((k_piece<4, true>), grid, dim3(256), 0, 0, buf, S, P, npiece, ngroups, order, sink, S2, nkq); else hipLaunchKernelGGL((k_piece<4, false>), grid, dim3(256), 0, 0, buf, S, P, npiece, ngroups, order, sink, S2, nkq); }
        else { if (wr) hipLaunchKernelGGL((k_piece<16, true>), grid, dim3(256), 0, 0, buf, S, P, npiece, ngroups, order, sink, S2, nkq); else hipLaunchKernelGGL((k_piece<16, false>), grid, dim3(256), 0, 0, buf, S, P, npiece, ngroups, order, sink, S2, nkq); }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (it && ms < best) best = ms;
    }
    const double bytes = (double)rows * span;
    printf("%s P=%5d S=%8zu span=%6d V=%2d order=%d nk=%d rows=%zu: %.3f ms  %.2f TB/s\n", wr ? "write" : "read ", P, S, span, V, order, nk, rows, best, bytes / best / 1e9);
    return 0;
}
