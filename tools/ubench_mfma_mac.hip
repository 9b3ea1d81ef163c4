// Probe: the MAC tile on v_mfma_f64_16x16x4_f64 instead of v_fmac_f64_dpp (same exact-fp64 limb arithmetic; NOT what the product path uses - the north star keeps
// MFMA to the plaintext projection).  A wave = one coefficient x 30 (32) rows x 48 columns x 3 limbs: per k-block of 4 k-steps 2 A operands (rot rows 0..15, 16..31),
// 3 plaintext words (-> 9 limb doubles by v_perm_b32) and 18 MFMAs (18 432 FMAs); 72 accumulator doubles per lane.
//   part 1: operands in registers;  part 2: operands from LDS each k-block (no DMA, no barrier), 8 waves per CU
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

template <int LDSFED>
__global__ void __launch_bounds__(512, 2) k_mfma(double *out, int iters, double seed) {
    extern __shared__ double lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    d4 acc[2][3][3];
    for (int r = 0; r < 2; r++) for (int c = 0; c < 3; c++) for (int l = 0; l < 3; l++) acc[r][c][l] = (d4){0.0, 0.0, 0.0, 0.0};
    if (LDSFED) { for (int i = tid; i < 8 * 1024; i += 512) lds[i] = seed + i; __syncthreads(); }
    double a0 = seed + lane, a1 = seed * 3 + lane;
    double pl[3][3];
    for (int c = 0; c < 3; c++) for (int l = 0; l < 3; l++) pl[c][l] = seed + c + l;
    const double *wl = lds + wave * 1024;
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        if (LDSFED) {
            const int kb = (it & 7) * 128;
            a0 = wl[kb + lane]; a1 = wl[kb + 64 + lane];
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const u64 p = (u64)__double_as_longlong(wl[(kb + c * 64 + lane) & 1023]);
                const unsigned plo = (unsigned)p, phi = (unsigned)(p >> 32);
                pl[c][0] = __hiloint2double((int)__builtin_amdgcn_perm(plo, 0u, 0x0C05040Cu), 0);
                pl[c][1] = __hiloint2double((int)__builtin_amdgcn_perm(plo, 0u, 0x0C07060Cu), 0);
                pl[c][2] = __hiloint2double((int)__builtin_amdgcn_perm(phi, 0u, 0x0C05040Cu), 0);
            }
        }
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int l = 0; l < 3; l++) {
                acc[0][c][l] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, pl[c][l], acc[0][c][l], 0, 0, 0);
                acc[1][c][l] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, pl[c][l], acc[1][c][l], 0, 0, 0);
            }
        if (!LDSFED) asm volatile("" : "+v"(a0), "+v"(a1));
    }
    double s = 0;
    for (int r = 0; r < 2; r++) for (int c = 0; c < 3; c++) for (int l = 0; l < 3; l++) s += acc[r][c][l][0] + acc[r][c][l][1] + acc[r][c][l][2] + acc[r][c][l][3];
    out[blockIdx.x * 512 + tid] = s;
}

int main() {
    double *out; hipMalloc(&out, 256 * 512 * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4000;
    for (int fed = 0; fed < 2; fed++) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0);
            if (fed) hipLaunchKernelGGL(k_mfma<1>, dim3(256), dim3(512), 8 * 1024 * 8, 0, out, iters, 1e-300);
            else hipLaunchKernelGGL(k_mfma<0>, dim3(256), dim3(512), 0, 0, out, iters, 1.5);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double fma = 256.0 * 8 * iters * 18 * 1024;
            if (rep) printf("%-58s %8.3f ms  %.3e FMA/s  %.2f FMA/clk/SIMD@2.4GHz\n", fed ? "MFMA f64 16x16x4 MAC tile, LDS-fed (8 waves per CU)" : "MFMA f64 16x16x4 MAC tile, operands in registers", ms, fma / (ms * 1e-3), fma / (ms * 1e-3) / (1024.0 * 2.4e9));
        }
    }
    return 0;
}
