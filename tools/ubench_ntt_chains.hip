// would TWO independent (plaintext, modulus) rows interleaved in one instruction stream (256 threads, <= 256 VGPRs, 2 workgroups per CU) issue the
// plaintext NTT's butterflies faster than ONE row per workgroup (128 VGPRs, 4 workgroups per CU)?  The probe is the NTT's phase B as the library runs it - the same
// ct_stage / mulmod_lazy_q code (sfgwas_amd/csrc/ntt_core.hpp), 32 values per row and thread, five stages with per-thread twiddles from a global table, then an exchange
// through LDS with a workgroup barrier - repeated ITER times, with ROWS = 1 or 2 rows per thread.  Output: butterflies per second and the fp64 issue-slot fraction
// (8 fp64 instructions per butterfly against 256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I sfgwas_amd/csrc tools/ubench_ntt_chains.hip -o tools/ubench_ntt_chains
#include "common.hpp"
#include "ntt_core.hpp"
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
constexpr int ITER = 64;
constexpr int ROWD = 16 * 16 + 8;             // doubles per `a` row of the half exchange image (padded): 16 x 264 x 8 B = 33 KiB per row

template <int ROWS, int WG_PER_CU>
__global__ void __launch_bounds__(256, WG_PER_CU) k_phase(const double *tw, double *out, double q, double qinv) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, a_b = tid >> 4, c_b = tid & 15;
    double v[ROWS][32];
#pragma unroll
    for (int r = 0; r < ROWS; r++)
#pragma unroll
        for (int b = 0; b < 32; b++) v[r][b] = (double)((tid * 37 + b * 11 + r * 5 + blockIdx.x) & 1023);
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int r = 0; r < ROWS; r++) {          // (the compiler is free to interleave the rows' butterflies: they share nothing but q)
            const int ab = a_b + 16 * r;
            ct_stage<32, 16>(v[r], q, qinv, [&](int g) { return tw[32 + ab + g]; });
            ct_stage<32, 8>(v[r], q, qinv, [&](int g) { return tw[64 + ab * 2 + g]; });
            ct_stage<32, 4>(v[r], q, qinv, [&](int g) { return tw[128 + ab * 4 + g]; });
            ct_stage<32, 2>(v[r], q, qinv, [&](int g) { return tw[256 + ab * 8 + g]; });
            ct_stage<32, 1>(v[r], q, qinv, [&](int g) { return tw[512 + ab * 16 + g]; });
        }
        // exchange through a HALF image per row (33 KiB, as k_ntt_half3's A->B exchange: two rounds of 16 values), one workgroup barrier each way and round
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int r = 0; r < ROWS; r++)
#pragma unroll
                for (int b = 0; b < 16; b++) lds[r * 16 * ROWD + a_b * ROWD + b * 16 + c_b] = v[r][h * 16 + b];
            __syncthreads();
#pragma unroll
            for (int r = 0; r < ROWS; r++)
#pragma unroll
                for (int b = 0; b < 16; b++) v[r][h * 16 + b] = pred(lds[r * 16 * ROWD + a_b * ROWD + c_b * 16 + b], q, qinv) + 1.0;      // (kept small: the loop would otherwise grow the lazy values past 2^51)
            __syncthreads();
        }
    }
    double s = 0;
#pragma unroll
    for (int r = 0; r < ROWS; r++)
#pragma unroll
        for (int b = 0; b < 32; b++) s += v[r][b];
    if (s == 0.12345) out[0] = s;
}

template <int ROWS, int WG_PER_CU> static void run(const char *name, const double *tw, double *out, int rows_total) {
    const size_t lds = (size_t)ROWS * 16 * ROWD * 8;
    CK(hipFuncSetAttribute((const void *)k_phase<ROWS, WG_PER_CU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int blocks = rows_total / ROWS;
    const double q = 34359214081.0, qinv = 1.0 / q;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_phase<ROWS, WG_PER_CU>), dim3(blocks), dim3(256), lds, 0, tw, out, q, qinv);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_phase<ROWS, WG_PER_CU>), dim3(blocks), dim3(256), lds, 0, tw, out, q, qinv);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double bf = (double)rows_total * 256 * ITER * 5 * 16;                 // butterflies (lanes): rows x threads x iterations x stages x 16 per stage
    const double pred_ops = (double)rows_total * 256 * ITER * 32 * 3;         // the exchange's pred + 1: mul, rndne, fma (+ add)
    printf("%-44s %8.3f ms  %.3e butterflies/s  fp64 issue fraction %.3f (butterflies only) %.3f (with the exchange's reductions)\n", name, ms, bf / (ms * 1e-3),
           8.0 * bf / (ms * 1e-3) / (256.0 * 4 * 16 * 2.4e9), (8.0 * bf + pred_ops + (double)rows_total * 256 * ITER * 32) / (ms * 1e-3) / (256.0 * 4 * 16 * 2.4e9));
}

int main() {
    std::vector<double> tw(16384);
    for (size_t i = 0; i < tw.size(); i++) tw[i] = (double)((i * 2654435761ULL) % 34359214081ULL);
    double *dtw, *out; CK(hipMalloc(&dtw, tw.size() * 8)); CK(hipMalloc(&out, 64));
    CK(hipMemcpy(dtw, tw.data(), tw.size() * 8, hipMemcpyHostToDevice));
    for (int rounds : {5, 20}) {
        const int rows = 256 * 4 * rounds;                                       // `rounds` resident rounds of the one-row kernel
        printf("== %d rows (%d rounds of 4 one-row workgroups per CU)\n", rows, rounds);
        run<1, 4>("1 row per workgroup, 128 VGPRs, 4 WG/CU", dtw, out, rows);
        run<1, 2>("1 row per workgroup, 256 VGPRs, 2 WG/CU", dtw, out, rows);
        run<2, 2>("2 rows per workgroup, 256 VGPRs, 2 WG/CU", dtw, out, rows);
        run<2, 1>("2 rows per workgroup, 512 VGPRs, 1 WG/CU", dtw, out, rows);
    }
    return 0;
}
