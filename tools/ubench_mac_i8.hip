// Probe (not on the product path): the ring MAC as an int8 matrix-core GEMM per coefficient.
//   acc[n][r] += sum_k pt[k][n] * rot[k][r] mod q for ONE coefficient is a 32 x 96 x K GEMM.  Both operands are split into 5 signed base-256 digits; the 25 digit
//   products go through v_mfma_i32_16x16x64_i8 into 9 int32 accumulators per output (one per a + b: |sum| <= 5 K 2^14 < 2^31), recombined mod q in the epilogue.
//   A wave = one coefficient pair (c, N-1-c: the plaintext word is shared) x 16 columns: 4 row tiles x 9 diagonals = 36 accumulator tiles, operands streamed from
//   global memory in the exact register layout (1 KiB per operand tile and digit, k-contiguous) - which is what a transposition pass would have to produce from the
//   coefficient-contiguous panel.  Six such waves (96 columns) form a workgroup and share the rot operand through the cache.
// Prints: correctness of sampled outputs against 128-bit host arithmetic, ms per launch-equivalent, operand GB/s, int8 MAC/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
constexpr int NCH = 23, NJT = 6, ND = 5;          // K = 1472 = 16 block rows x 91 (+16 pad), 96 columns, 5 digits
__host__ __device__ inline unsigned hash32(u64 x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return (unsigned)x; }
// digit byte of operand stream `which` (0 = A / rot, 1 = B / pt) at byte offset `off`
__host__ __device__ inline int8_t digit(int which, u64 off) { return (int8_t)(hash32(off * 2 + which) & 0xFF); }
__global__ void k_fill(int8_t *p, u64 n, int which) { for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < n; i += (u64)gridDim.x * 256) p[i] = digit(which, i); }

// A: [c' < 2 NC][ch][rt 2][a 5][lane 64][16]   B: [c < NC][jt 6][ch][b 5][lane 64][16]   out: [c][half 2][jt][rt 2][lane 64][4] u64 (row-contiguous runs of 4)
__global__ void __launch_bounds__(384, 1) k_mac_i8(const uint4 *A, const uint4 *B, u64 *out, int NC, double q, double qinv, int accumulate) {
    const int lane = threadIdx.x & 63, jt = threadIdx.x >> 6, c = blockIdx.x;
    const int cbar = 2 * NC - 1 - c;
    v4i acc[4][9];
#pragma unroll
    for (int t = 0; t < 4; t++)
#pragma unroll
        for (int s = 0; s < 9; s++) acc[t][s] = (v4i){0, 0, 0, 0};
    const uint4 *Bp = B + ((size_t)(c * NJT + jt) * NCH) * ND * 64 + lane;
    const uint4 *A0 = A + ((size_t)c * NCH) * 2 * ND * 64 + lane, *A1 = A + ((size_t)cbar * NCH) * 2 * ND * 64 + lane;
#pragma unroll 1
    for (int ch = 0; ch < NCH; ch++) {
        v4i b[ND];
#pragma unroll
        for (int d = 0; d < ND; d++) { const uint4 w = Bp[(size_t)(ch * ND + d) * 64]; b[d] = (v4i){(int)w.x, (int)w.y, (int)w.z, (int)w.w}; }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const uint4 *Ap = (t < 2 ? A0 : A1) + (size_t)((ch * 2 + (t & 1)) * ND) * 64;
#pragma unroll
            for (int a = 0; a < ND; a++) {
                const uint4 w = Ap[(size_t)a * 64];
                const v4i av = (v4i){(int)w.x, (int)w.y, (int)w.z, (int)w.w};
#pragma unroll
                for (int d = 0; d < ND; d++) acc[t][a + d] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, b[d], acc[t][a + d], 0, 0, 0);
            }
        }
        __syncthreads();                                        // the six column waves stay within one chunk of each other: A is fetched once
    }
    // epilogue: value = sum_s D_s 256^s mod q by Horner (|r| < q, r 256 + D < 2^44: exact in fp64), then the accumulator read-modify-write
#pragma unroll
    for (int t = 0; t < 4; t++) {
        u64 *o = out + ((((size_t)c * 2 + (t >> 1)) * NJT + jt) * 2 + (t & 1)) * 256 + lane * 4;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            double r = (double)acc[t][8][e];
#pragma unroll
            for (int s = 7; s >= 0; s--) { const double x = r * 256.0 + (double)acc[t][s][e]; r = x - q * __builtin_rint(x * qinv); }
            if (accumulate) { const double x = r + (double)o[e]; r = x - q * __builtin_rint(x * qinv); }
            if (r < 0) r += q;
            o[e] = (u64)r;
        }
    }
}
int main() {
    const int NC = 2048;
    const double q = 34359214081.0;
    const size_t nA = (size_t)2 * NC * NCH * 2 * ND * 1024, nB = (size_t)NC * NJT * NCH * ND * 1024, nO = (size_t)NC * 2 * NJT * 2 * 256;
    int8_t *A, *B; u64 *out;
    hipMalloc(&A, nA); hipMalloc(&B, nB); hipMalloc(&out, nO * 8);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, A, (u64)nA, 0);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, B, (u64)nB, 1);
    hipMemset(out, 0, nO * 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_mac_i8, dim3(NC), dim3(384), 0, 0, (const uint4 *)A, (const uint4 *)B, out, NC, q, 1.0 / q, rep > 0);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    }
    // after 3 passes (1 overwrite + 2 accumulate) every output = 3 x the product sum mod q: check samples with 128-bit host arithmetic
    std::vector<u64> h(nO); hipMemcpy(h.data(), out, nO * 8, hipMemcpyDeviceToHost);
    const u64 qi = (u64)q; int bad = 0, checked = 0;
    for (int smp = 0; smp < 48; smp++) {
        const int c = (smp * 977) % NC, half = smp & 1, jt = smp % NJT, rt = (smp >> 1) & 1, lane = (smp * 13) % 64, e = smp & 3;
        const int cp = half ? 2 * NC - 1 - c : c;
        // accumulator layout of the 16x16 int32 tile: lane = col j + 16 * (row / 4), element = row % 4
        const int j = lane & 15, i = (lane >> 4) * 4 + e;
        __int128 sum = 0;
        for (int ch = 0; ch < NCH; ch++) for (int g = 0; g < 4; g++) for (int x = 0; x < 16; x++) {
            long long av = 0, bv = 0;
            for (int d = ND - 1; d >= 0; d--) {
                const u64 offA = ((((u64)cp * NCH + ch) * 2 + rt) * ND + d) * 1024 + (u64)(i + 16 * g) * 16 + x;
                const u64 offB = ((((u64)c * NJT + jt) * NCH + ch) * ND + d) * 1024 + (u64)(j + 16 * g) * 16 + x;
                av = av * 256 + digit(0, offA); bv = bv * 256 + digit(1, offB);
            }
            sum += (__int128)av * bv;
        }
        long long ref = (long long)(((sum % (__int128)qi) + qi) % qi); ref = (long long)(((__int128)ref * 3) % qi);
        const u64 got = h[((((size_t)c * 2 + half) * NJT + jt) * 2 + rt) * 256 + lane * 4 + e];
        checked++; if (got != (u64)ref) { if (bad < 4) printf("MISMATCH smp %d: got %llu ref %lld\n", smp, got, ref); bad++; }
    }
    const double macs = (double)NC * NJT * NCH * 100 * 16384.0, bytes = (double)nB + nA;
    printf("int8-MFMA ring MAC probe: %d coefficient pairs x 64 rows x 96 columns x K = %d, one 35-bit modulus\n", NC, NCH * 64);
    printf("  sampled outputs vs 128-bit host arithmetic: %d of %d equal\n", checked - bad, checked);
    printf("  %.3f ms   operand stream %.2f GB -> %.2f TB/s   %.3e int8 MAC/s (%.1f %% of 2.5e15)   %.3e ring-MAC/s (the fp64 DPP kernel: ~7.6e12 padded)\n",
           ms, bytes / 1e9, bytes / (ms * 1e-3) / 1e12, macs / (ms * 1e-3), 100 * macs / (ms * 1e-3) / 2.5e15, macs / 25 / (ms * 1e-3));
    printf("  scaled to one launch of the product (4 small moduli x 8192 pairs): %.2f ms\n", ms * 4 * 8192 / NC);
    return bad != 0;
}
