// FMA-rate probe with the register footprint of the MAC thread tile (8 rows x 3 cols x 3 limbs = 72 accumulators,
// 8 rot operands, 9 limb operands).  MODE 0: no memory traffic, three FMA orders.  MODE 1: operands re-read from LDS every
// k-step exactly as k_mac_dma does (8 rot doubles + 3 plaintext words + limb conversion, software-pipelined), no DMA, no barrier.
// MODE 2: as 1 but the limbs are read as doubles (no conversion).  MODE 3: as 1 with the rot reads only.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;
template <int ORDER>
__global__ void __launch_bounds__(512, 2) k_tile(double *out, int iters, double seed) {
    double acc[8][3][3], rr[8], pp[3][3];
    for (int r = 0; r < 8; r++) { rr[r] = seed + r + threadIdx.x; for (int t = 0; t < 3; t++) for (int l = 0; l < 3; l++) acc[r][t][l] = 0.0; }
    for (int t = 0; t < 3; t++) for (int l = 0; l < 3; l++) pp[t][l] = seed * (t + 2) + l;
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        if (ORDER == 0) {
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int t = 0; t < 3; t++)
#pragma unroll
                    for (int l = 0; l < 3; l++) acc[r][t][l] = __builtin_fma(rr[r], pp[t][l], acc[r][t][l]);
        } else if (ORDER == 1) {
#pragma unroll
            for (int t = 0; t < 3; t++)
#pragma unroll
                for (int l = 0; l < 3; l++)
#pragma unroll
                    for (int r = 0; r < 8; r++) acc[r][t][l] = __builtin_fma(rr[r], pp[t][l], acc[r][t][l]);
        } else {
#pragma unroll
            for (int l = 0; l < 3; l++)
#pragma unroll
                for (int r = 0; r < 8; r++)
#pragma unroll
                    for (int t = 0; t < 3; t++) acc[r][t][l] = __builtin_fma(rr[r], pp[t][l], acc[r][t][l]);
        }
#pragma unroll
        for (int r = 0; r < 8; r++) asm volatile("" : "+v"(rr[r]));
#pragma unroll
        for (int t = 0; t < 3; t++)
#pragma unroll
            for (int l = 0; l < 3; l++) asm volatile("" : "+v"(pp[t][l]));
    }
    double s = 0;
    for (int r = 0; r < 8; r++) for (int t = 0; t < 3; t++) for (int l = 0; l < 3; l++) s += acc[r][t][l];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
__global__ void __launch_bounds__(512, 2) k_tile_lds(double *out, int iters, double seed) {
    extern __shared__ double lds[];                      // rot [4][32][16] doubles (16 KiB), then pt [4][24][16] words or 3 limb planes
    double *rot = lds; u64 *ptw = reinterpret_cast<u64 *>(lds + 2048); double *ptl = lds + 2048;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cc = lane & 15, cg = lane >> 4, rh = wave & 3, wc = wave >> 2;
    for (int i = tid; i < 2048; i += 512) rot[i] = seed + i;
    for (int i = tid; i < 3 * 1536; i += 512) { if (MODE == 2) ptl[i] = (double)(i & 4095); else if (i < 1536) ptw[i] = 0x123456789ULL + i; }
    __syncthreads();
    double acc[8][3][3];
    for (int r = 0; r < 8; r++) for (int t = 0; t < 3; t++) for (int l = 0; l < 3; l++) acc[r][t][l] = 0.0;
    double rcur[8], rnxt[8]; u64 wcur[3], wnxt[3]; double lcur[3][3], lnxt[3][3];
    auto fetch = [&](int kk, double (&rr)[8], u64 (&ww)[3], double (&ll)[3][3]) {
#pragma unroll
        for (int t = 0; t < 3; t++) {
            const int w = (kk * 24 + (wc * 4 + cg) * 3 + t) * 16 + cc;
            if (MODE == 2) { ll[t][0] = ptl[w]; ll[t][1] = ptl[1536 + w]; ll[t][2] = ptl[3072 + w]; }
            else if (MODE == 1) ww[t] = ptw[w];
        }
#pragma unroll
        for (int r = 0; r < 8; r++) rr[r] = rot[(kk * 32 + rh * 8 + r) * 16 + cc];
    };
    auto fmas = [&](const double (&rr)[8], const u64 (&ww)[3], const double (&ll)[3][3]) {
        double p[3][3];
#pragma unroll
        for (int t = 0; t < 3; t++) {
            if (MODE == 1) {
                const unsigned plo = (unsigned)ww[t], phi = (unsigned)(ww[t] >> 32);
                p[t][0] = (double)(plo & 0xFFFu); p[t][1] = (double)((plo >> 12) & 0xFFFu); p[t][2] = (double)((plo >> 24) | (phi << 8));
            } else if (MODE == 2) { p[t][0] = ll[t][0]; p[t][1] = ll[t][1]; p[t][2] = ll[t][2]; }
            else { p[t][0] = seed; p[t][1] = seed + 1; p[t][2] = seed + 2; }
        }
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int t = 0; t < 3; t++)
#pragma unroll
                for (int l = 0; l < 3; l++) acc[r][t][l] = __builtin_fma(rr[r], p[t][l], acc[r][t][l]);
    };
#pragma unroll 1
    for (int it = 0; it < iters; it += 4) {
        fetch(0, rcur, wcur, lcur);
        fetch(1, rnxt, wnxt, lnxt); fmas(rcur, wcur, lcur);
        fetch(2, rcur, wcur, lcur); fmas(rnxt, wnxt, lnxt);
        fetch(3, rnxt, wnxt, lnxt); fmas(rcur, wcur, lcur);
        fmas(rnxt, wnxt, lnxt);
        asm volatile("" ::: "memory");
    }
    double s = 0;
    for (int r = 0; r < 8; r++) for (int t = 0; t < 3; t++) for (int l = 0; l < 3; l++) s += acc[r][t][l];
    out[blockIdx.x * 512 + tid] = s;
}

template <class K> void run(const char *name, K kern, int nblk, size_t ldsb) {
    double *out; (void)hipMalloc(&out, (size_t)nblk * 512 * 8);
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(512), ldsb, 0, out, 100, 1.5);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(nblk), dim3(512), ldsb, 0, out, iters, 1.5); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double fma = (double)nblk * 512 * 72.0 * iters;
    printf("%-44s blocks=%d  %.3f ms  %.3e FMA/s  %.2f lanes/clk/SIMD@2.4GHz\n", name, nblk, ms, fma / (ms * 1e-3), fma / (ms * 1e-3) / (1024 * 2.4e9));
    (void)hipFree(out);
}
int main() {
    const size_t L = 128 * 1024;    // same LDS footprint as the MAC kernel: one workgroup per CU
    (void)hipFuncSetAttribute((const void *)k_tile_lds<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L);
    (void)hipFuncSetAttribute((const void *)k_tile_lds<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L);
    (void)hipFuncSetAttribute((const void *)k_tile_lds<3>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L);
    run("no memory: r,t,l (MAC kernel order)", k_tile<0>, 256, 0);
    run("no memory: t,l,r", k_tile<1>, 256, 0);
    run("no memory: l,r,t", k_tile<2>, 256, 0);
    run("LDS: 8 rot + 3 words + limb conversion", k_tile_lds<1>, 256, L);
    run("LDS: 8 rot + 9 limb doubles", k_tile_lds<2>, 256, L);
    run("LDS: 8 rot only", k_tile_lds<3>, 256, L);
    return 0;
}
