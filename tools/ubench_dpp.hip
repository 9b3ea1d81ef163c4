// Probe for the DPP-broadcast MAC tile: v_fmac_f64_dpp with row_newbcast (the one DPP control the fp64 ALU accepts on gfx90a+).
// Tile under test: lane = (coefficient rho < 4 = DPP row, column i < 16 = lane in row); lane i of a DPP row ALSO holds the rot
// operand of rows i and 16+i for that coefficient, and every FMA takes its rot operand by row_newbcast from the lane that holds it.
// Thread tile = 30 rows x 1 column x 3 limbs = 90 accumulators; per k-step a thread reads 2 rot doubles + 1 plaintext word from LDS.
//   part 0: semantics check of row_newbcast
//   part 1: FMA rate, operands in registers: plain v_fma_f64 vs v_fmac_f64_dpp
//   part 2: FMA rate with the per-k-step LDS reads + limb unpack (v_perm_b32) of the real loop, software-pipelined, no DMA/barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned long long u64;

#define FMAC_DPP(ACC, ROT, P, LANE) asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #LANE " row_mask:0xf bank_mask:0xf" : "+v"(ACC) : "v"(ROT), "v"(P))
// rows 0..14 from register A (lanes 0..14), rows 15..29 from register B (lanes 0..14)
#define ROWS15(M, ACC, ROT, P, L) \
    M(ACC[0][L], ROT, P, 0); M(ACC[1][L], ROT, P, 1); M(ACC[2][L], ROT, P, 2); M(ACC[3][L], ROT, P, 3); M(ACC[4][L], ROT, P, 4); \
    M(ACC[5][L], ROT, P, 5); M(ACC[6][L], ROT, P, 6); M(ACC[7][L], ROT, P, 7); M(ACC[8][L], ROT, P, 8); M(ACC[9][L], ROT, P, 9); \
    M(ACC[10][L], ROT, P, 10); M(ACC[11][L], ROT, P, 11); M(ACC[12][L], ROT, P, 12); M(ACC[13][L], ROT, P, 13); M(ACC[14][L], ROT, P, 14)

__global__ void k_sem(double *out, const double *in) {
    double acc = 0.0, r = in[threadIdx.x], p = 1.0;
    FMAC_DPP(acc, r, p, 5);
    out[threadIdx.x] = acc;          // expect in[(lane & ~15) + 5]
}

// subnormal multiplier semantics: the double with exponent field 0 and mantissa bits 51..40 = x is x * 2^-1034; products with integers stay exact and
// two multiplications by 2^517 bring the sum back
__global__ void k_subnormal(double *out) {
    const unsigned x = 0xABC + threadIdx.x;
    double lim = __hiloint2double((int)(x << 8), 0), acc = 0.0, r = 12345678901.0 + threadIdx.x;
    for (int k = 0; k < 100; k++) acc = __builtin_fma(r, lim, acc);
    out[threadIdx.x] = (acc * 0x1p517) * 0x1p517 - 100.0 * r * (double)x;          // expect 0
}
template <int DPP>
__global__ void __launch_bounds__(512, 2) k_reg(double *out, int iters, double seed) {
    double a0[15][3], a1[15][3], ra = seed + threadIdx.x, rb = seed * 3 + threadIdx.x, p[3] = {seed, seed < 1e-300 ? seed * 2 : seed + 1, seed < 1e-300 ? seed * 3 : seed + 2};
    for (int r = 0; r < 15; r++) for (int l = 0; l < 3; l++) a0[r][l] = a1[r][l] = 0.0;
#pragma unroll 1
    for (int it = 0; it < iters; it++) {
        if (DPP) {
#pragma unroll
            for (int l = 0; l < 3; l++) { ROWS15(FMAC_DPP, a0, ra, p[l], l); ROWS15(FMAC_DPP, a1, rb, p[l], l); }
        } else {
#pragma unroll
            for (int l = 0; l < 3; l++)
#pragma unroll
                for (int r = 0; r < 15; r++) { a0[r][l] = __builtin_fma(ra, p[l], a0[r][l]); a1[r][l] = __builtin_fma(rb, p[l], a1[r][l]); }
        }
        asm volatile("" : "+v"(ra), "+v"(rb), "+v"(p[0]), "+v"(p[1]), "+v"(p[2]));
    }
    double s = 0;
    for (int r = 0; r < 15; r++) for (int l = 0; l < 3; l++) s += a0[r][l] + a1[r][l];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

// LDS-fed: rot [4 k][32 rows][16 coefficients] doubles, pt [4 k][32 cols][16 coefficients] packed-limb words; the XOR swizzle of the
// 16-byte granules inside each 8-row group that the DMA would apply is modelled in the read addresses
template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES, 2) k_lds(double *out, int iters, double seed) {
    extern __shared__ double lds[];
    double *rot = lds; u64 *ptw = reinterpret_cast<u64 *>(lds + 4 * 32 * 16);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, rho = lane >> 4, i = lane & 15, cq = wave & 3, ch = wave >> 2;
    for (int x = tid; x < 4 * 32 * 16; x += 64 * WAVES) { rot[x] = seed + x; ptw[x] = 0xB123B456B789ULL + x; }
    __syncthreads();
    double a0[15][3], a1[15][3];
    for (int r = 0; r < 15; r++) for (int l = 0; l < 3; l++) a0[r][l] = a1[r][l] = 0.0;
    const int cc = cq * 4 + rho;
    auto sw = [&](int row, int c) { return (row >> 3) * 128 + (row & 7) * 16 + ((((c >> 1) ^ (row & 7)) << 1) | (c & 1)); };   // doubles inside one k-step image
    const int ra_off = sw(i, cc), rb_off = sw(16 + i < 30 ? 16 + i : 29, cc), p_off = sw((ch & 1) * 16 + i, cc);
    double pl[3];
    for (int k = 0; k < 3; k++) { int z; asm volatile("v_mov_b32 %0, 0" : "=v"(z)); pl[k] = __hiloint2double(0, z); }
    double ra_c, rb_c, ra_n, rb_n; u64 p_c, p_n;
    auto fetch = [&](int kk, double &ra, double &rb, u64 &p) { ra = rot[kk * 512 + ra_off]; rb = rot[kk * 512 + rb_off]; p = ptw[kk * 512 + p_off]; };
    auto fmas = [&](double ra, double rb, u64 p) {
        const unsigned plo = (unsigned)p, phi = (unsigned)(p >> 32);
        pl[0] = __hiloint2double((int)__builtin_amdgcn_perm(plo, 0x40404040u, 0x0005040Cu), __double2loint(pl[0]));
        pl[1] = __hiloint2double((int)__builtin_amdgcn_perm(plo, 0x40404040u, 0x0007060Cu), __double2loint(pl[1]));
        pl[2] = __hiloint2double((int)__builtin_amdgcn_perm(phi, 0x40404040u, 0x0005040Cu), __double2loint(pl[2]));
#pragma unroll
        for (int l = 0; l < 3; l++) { ROWS15(FMAC_DPP, a0, ra, pl[l], l); ROWS15(FMAC_DPP, a1, rb, pl[l], l); }
    };
#pragma unroll 1
    for (int it = 0; it < iters; it += 4) {
        fetch(0, ra_c, rb_c, p_c);
        fetch(1, ra_n, rb_n, p_n); fmas(ra_c, rb_c, p_c);
        fetch(2, ra_c, rb_c, p_c); fmas(ra_n, rb_n, p_n);
        fetch(3, ra_n, rb_n, p_n); fmas(ra_c, rb_c, p_c);
        fmas(ra_n, rb_n, p_n);
        asm volatile("" ::: "memory");
    }
    double s = 0;
    for (int r = 0; r < 15; r++) for (int l = 0; l < 3; l++) s += a0[r][l] + a1[r][l];
    out[blockIdx.x * 64 * WAVES + tid] = s;
}

template <class K> void run(const char *name, K kern, int nblk, int threads, size_t ldsb) {
    double *out; (void)hipMalloc(&out, (size_t)nblk * threads * 8);
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(threads), ldsb, 0, out, 100, 1.5);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(nblk), dim3(threads), ldsb, 0, out, iters, 1.5); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double fma = (double)nblk * threads * 90.0 * iters;
    printf("%-60s blocks=%d x %d  %.3f ms  %.3e FMA/s  %.2f lanes/clk/SIMD@2.4GHz\n", name, nblk, threads, ms, fma / (ms * 1e-3), fma / (ms * 1e-3) / (1024 * 2.4e9));
    (void)hipFree(out);
}
template <class K> void run_seed(const char *name, K kern, int nblk, int threads, size_t ldsb, double seed) {
    double *out; (void)hipMalloc(&out, (size_t)nblk * threads * 8);
    const int iters = 20000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(nblk), dim3(threads), ldsb, 0, out, 100, seed);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(nblk), dim3(threads), ldsb, 0, out, iters, seed); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    double fma = (double)nblk * threads * 90.0 * iters, h; (void)hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost);
    printf("%-60s blocks=%d x %d  %.3f ms  %.3e FMA/s  %.2f lanes/clk/SIMD@2.4GHz   (sample result %.6g)\n", name, nblk, threads, ms, fma / (ms * 1e-3), fma / (ms * 1e-3) / (1024 * 2.4e9), h);
    (void)hipFree(out);
}
int main() {
    {   // semantics
        double h[64], *din, *dout, o[64]; for (int i = 0; i < 64; i++) h[i] = 100.0 + i;
        (void)hipMalloc(&din, 512); (void)hipMalloc(&dout, 512); (void)hipMemcpy(din, h, 512, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, dout, din); (void)hipMemcpy(o, dout, 512, hipMemcpyDeviceToHost);
        int bad = 0; for (int i = 0; i < 64; i++) if (o[i] != h[(i & ~15) + 5]) bad++;
        printf("row_newbcast:5 semantics: %s (lane 0 -> %.0f, lane 17 -> %.0f, lane 63 -> %.0f)\n", bad ? "UNEXPECTED" : "lane 5 of each 16-lane row, as assumed", o[0], o[17], o[63]);
    }
    {
        double *d, o[64]; (void)hipMalloc(&d, 512); hipLaunchKernelGGL(k_subnormal, dim3(1), dim3(64), 0, 0, d); (void)hipMemcpy(o, d, 512, hipMemcpyDeviceToHost);
        int bad = 0; for (int i = 0; i < 64; i++) if (o[i] != 0.0) bad++;
        printf("subnormal limb x * 2^-1034: 100-term sums %s\n", bad ? "NOT exact (flushed?)" : "exact after scaling back by 2^517 twice");
    }
    const size_t L64 = 64 * 1024, L128 = 128 * 1024;
    (void)hipFuncSetAttribute((const void *)k_lds<8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L128);
    (void)hipFuncSetAttribute((const void *)k_lds<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)L64);
    run("registers: plain v_fma_f64, 90 acc", k_reg<0>, 256, 512, 0);
    run("registers: v_fmac_f64_dpp row_newbcast, 90 acc", k_reg<1>, 256, 512, 0);
    run_seed("registers: v_fmac_f64_dpp, SUBNORMAL multiplier (3e-311)", k_reg<1>, 256, 512, 0, 3e-311);
    run("LDS-fed DPP tile, 8-wave WG, 1 per CU (2 waves/SIMD)", k_lds<8>, 256, 512, L128);
    run("LDS-fed DPP tile, 4-wave WG, 2 per CU (2 waves/SIMD)", k_lds<4>, 512, 256, L64);
    return 0;
}
