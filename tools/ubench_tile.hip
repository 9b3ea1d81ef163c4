// FMA-rate probe with the register footprint of the MAC thread tile (8 rows x 3 cols x 3 limbs = 72 accumulators,
// 8 rot operands, 9 limb operands), no memory traffic.  Variants differ only in the order of the 72 FMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
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
        // keep the operands changing so that nothing is hoisted
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
template <int ORDER> void run(const char *name, int nblk) {
    double *out; hipMalloc(&out, (size_t)nblk * 512 * 8);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_tile<ORDER>, dim3(nblk), dim3(512), 0, 0, out, 100, 1.5);
    hipEventRecord(e0); hipLaunchKernelGGL(k_tile<ORDER>, dim3(nblk), dim3(512), 0, 0, out, iters, 1.5); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double fma = (double)nblk * 512 * 72.0 * iters;
    printf("%-28s blocks=%d  %.3f ms  %.3e FMA/s  %.2f lanes/clk/SIMD@2.4GHz\n", name, nblk, ms, fma / (ms * 1e-3), fma / (ms * 1e-3) / (1024 * 2.4e9));
    hipFree(out);
}
int main() {
    for (int nblk : {256, 512}) { run<0>("r,t,l (MAC kernel order)", nblk); run<1>("t,l,r", nblk); run<2>("l,r,t", nblk); }
    return 0;
}
