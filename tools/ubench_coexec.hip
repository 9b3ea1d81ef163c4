// Probe: do v_mfma_f64_16x16x4_f64 and v_fma_f64 execute concurrently on gfx950?  A workgroup = 8 waves = 2 per SIMD (waves w and w + 4 share SIMD w % 4).
//   A: waves 0-3 run an MFMA f64 loop, waves 4-7 exit          B: waves 0-3 exit, waves 4-7 run an independent-chain v_fma_f64 loop
//   C: both (one MFMA wave + one VALU wave per SIMD)            D: every wave issues both kinds, interleaved by the compiler (1 MFMA : 16 v_fma)
// T_C ~ max(T_A, T_B): separate fp64 datapaths.  T_C ~ T_A + T_B: one datapath.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ void __launch_bounds__(512, 2) k(double *out, int iters, double seed) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool do_mfma = MODE == 3 || ((MODE == 0 || MODE == 2) && wave < 4);
    const bool do_valu = MODE == 3 || ((MODE == 1 || MODE == 2) && wave >= 4);
    double s = 0;
    if (MODE == 3) {
        d4 acc[6]; double v[32];
        for (int i = 0; i < 6; i++) acc[i] = (d4){0, 0, 0, 0};
        for (int i = 0; i < 32; i++) v[i] = seed * i;
        double a = seed + lane, b = seed * 2 + lane, m = 1.0 + seed;
#pragma unroll 1
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int j = 0; j < 6; j++) {
                acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 16; i++) { const int x = (j * 16 + i) & 31; v[x] = __builtin_fma(v[x], m, a); }
            }
            asm volatile("" : "+v"(a), "+v"(b));
        }
        for (int i = 0; i < 6; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        for (int i = 0; i < 32; i++) s += v[i];
    } else if (do_mfma) {
        d4 acc[18];
        for (int i = 0; i < 18; i++) acc[i] = (d4){0, 0, 0, 0};
        double a = seed + lane, b = seed * 2 + lane;
#pragma unroll 1
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int j = 0; j < 18; j++) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
            asm volatile("" : "+v"(a), "+v"(b));
        }
        for (int i = 0; i < 18; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else if (do_valu) {
        double v[36];
        for (int i = 0; i < 36; i++) v[i] = seed * i;
        double a = seed + lane, m = 1.0 + seed;
#pragma unroll 1
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int r = 0; r < 8; r++)
#pragma unroll
                for (int i = 0; i < 36; i++) v[i] = __builtin_fma(v[i], m, a);
            asm volatile("" : "+v"(a), "+v"(m));
        }
        for (int i = 0; i < 36; i++) s += v[i];
    } else return;
    out[blockIdx.x * 512 + tid] = s;
}

template <int MODE> static float run(double *out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, out, iters, 1e-30);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    return ms;
}
int main() {
    double *out; hipMalloc(&out, 256 * 512 * 8);
    const int iters = 4000;
    const double clk = 2.4e9, simds = 1024.0;
    const float tA = run<0>(out, iters), tB = run<1>(out, iters), tC = run<2>(out, iters), tD = run<3>(out, iters);
    const double fA = simds * iters * 18 * 1024.0, fB = simds * iters * 288 * 64.0, fD = simds * 2 * iters * (6 * 1024.0 + 96 * 64.0);
    printf("A  MFMA f64 only (1 wave/SIMD):            %8.3f ms  %.2f FMA/clk/SIMD\n", tA, fA / (tA * 1e-3) / (simds * clk));
    printf("B  v_fma_f64 only (1 wave/SIMD):           %8.3f ms  %.2f FMA/clk/SIMD\n", tB, fB / (tB * 1e-3) / (simds * clk));
    printf("C  one MFMA wave + one VALU wave per SIMD: %8.3f ms  %.2f FMA/clk/SIMD   (max(A,B) = %.3f, A+B = %.3f)\n", tC, (fA + fB) / (tC * 1e-3) / (simds * clk), tA > tB ? tA : tB, tA + tB);
    printf("D  both kinds in every wave (2 waves/SIMD): %8.3f ms  %.2f FMA/clk/SIMD\n", tD, fD / (tD * 1e-3) / (simds * clk));
    return 0;
}
