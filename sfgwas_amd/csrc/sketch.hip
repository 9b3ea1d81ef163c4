// sketch.hip — the plaintext pass of randomized PCA over the local genotypes (gwas/pca.go:152-162):
//     localSketch[bucket[i]][j] += sgn[i] * x[i][j]   (fp64),   xsum[j] += uint64(x),   x2sum[j] += uint64(x*x)
// i.e. sketch = S (kp x n_ind, one +-1 per column) times X (n_ind x m_snp).  This is the one place on the path
// where the matrix cores are used: the count-sketch projection is an exact small-integer GEMM, done with
// v_mfma_f64_16x16x4_f64 (A = 16 buckets x 4 individuals of S, B = 4 individuals x 16 SNPs of X converted from
// int8).  All values are integers far below 2^53, so the fp64 result is exact and order-independent.
// One 256-thread workgroup owns a 256-column strip of X over ALL rows (no atomics); genotype tiles of
// 64 rows x 256 B are staged through LDS with full-line coalesced reads.  The column moments are done on the
// vector ALU from the same tile with Go's integer semantics (uint64(int8) sign-extends; x*x wraps in int8).
#include "common.hpp"
#include "kernels.hpp"

typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int SK_ROWS = 64, SK_COLS = 256;

__global__ void __launch_bounds__(256) k_sketch(const int8_t *X, size_t nrow, size_t ncol, size_t ld, const int32_t *bucket, const int8_t *sgn,
                                                int kp, double *sketch, u64 *xsum, u64 *x2sum) {
    __shared__ int8_t tile[SK_ROWS][SK_COLS];
    __shared__ int32_t bk[SK_ROWS];
    __shared__ int8_t sg[SK_ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t col0 = (size_t)blockIdx.x * SK_COLS;
    d4 acc[4];                                   // 4 column groups of 16 per wave: 16 buckets x 16 SNPs each
#pragma unroll
    for (int g = 0; g < 4; g++) acc[g] = (d4){0.0, 0.0, 0.0, 0.0};
    u64 s1 = 0, s2 = 0;
    for (size_t r0 = 0; r0 < nrow; r0 += SK_ROWS) {
        __syncthreads();
        // stage 64 rows x 256 B: 16 B per thread per pass, rows are contiguous 256-B segments
        for (int e = tid; e < SK_ROWS * (SK_COLS / 16); e += 256) {
            const int rr = e / (SK_COLS / 16), cq = e % (SK_COLS / 16);
            const size_t row = r0 + rr, col = col0 + (size_t)cq * 16;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row < nrow) {
                if (col + 16 <= ncol && (reinterpret_cast<uintptr_t>(X + row * ld + col) & 15) == 0) v = *reinterpret_cast<const uint4 *>(X + row * ld + col);
                else { int8_t b[16]; for (int k = 0; k < 16; k++) b[k] = col + k < ncol ? X[row * ld + col + k] : (int8_t)0; v = *reinterpret_cast<uint4 *>(b); }
            }
            *reinterpret_cast<uint4 *>(&tile[rr][cq * 16]) = v;
        }
        if (tid < SK_ROWS) { const size_t row = r0 + tid; bk[tid] = row < nrow ? bucket[row] : -1; sg[tid] = row < nrow ? sgn[row] : (int8_t)0; }
        __syncthreads();
        // column moments: thread = column
        {
#pragma unroll 8
            for (int rr = 0; rr < SK_ROWS; rr++) {
                const int8_t x = tile[rr][tid];
                s1 += (u64)(long long)x;                          // uint64(row[j])      (pca.go:158)
                s2 += (u64)(long long)(int8_t)(x * x);            // uint64(row[j]*row[j]) in int8 arithmetic (:159)
            }
        }
        // projection on the matrix cores: lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]
        const int ai = lane & 15, ak = lane >> 4;
        for (int k0 = 0; k0 < SK_ROWS; k0 += 4) {
            const int rr = k0 + ak;
            const double a = bk[rr] == ai ? (double)sg[rr] : 0.0;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const double b = (double)tile[rr][wave * 64 + g * 16 + ai];
                acc[g] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[g], 0, 0, 0);
            }
        }
    }
    if (col0 + tid < ncol) { xsum[col0 + tid] = s1; x2sum[col0 + tid] = s2; }
    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int g = 0; g < 4; g++) {
        const size_t col = col0 + (size_t)wave * 64 + g * 16 + (lane & 15);
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = (lane >> 4) + 4 * r;
            if (row < kp && col < ncol) sketch[(size_t)row * ncol + col] = acc[g][r];
        }
    }
}

extern "C" int sfg_sketch(sfg_ctx *ctx, const sfg_geno *g, const int32_t *bucket_host, const int8_t *sgn_host, int kp,
                          double *sketch_host, uint64_t *xsum_host, uint64_t *x2sum_host) {
    if (g->packed) { ctx->err = "sfg_sketch: 2-bit packed matrix (sketch before sfg_geno_pack, or sfg_geno_unpack first)"; return 1; }
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (kp < 1 || kp > 16) SFG_FAIL(ctx, "sfg_sketch: kp must be in 1..16 (one MFMA tile of buckets)");
    for (size_t i = 0; i < g->nrow; i++) if (bucket_host[i] < 0 || bucket_host[i] >= kp) SFG_FAIL(ctx, "sfg_sketch: bucket index out of range");
    const size_t nrow = g->nrow, ncol = g->ncol;
    char *d = nullptr;
    const size_t b_bk = nrow * 4, b_sg = (nrow + 7) & ~(size_t)7, b_sk = (size_t)kp * ncol * 8, b_s = ncol * 8;
    SFG_HIP(ctx, hipMalloc(&d, b_bk + b_sg + b_sk + 2 * b_s));
    int32_t *dbk = (int32_t *)d; int8_t *dsg = (int8_t *)(d + b_bk); double *dsk = (double *)(d + b_bk + b_sg);
    u64 *dx = (u64 *)(d + b_bk + b_sg + b_sk), *dx2 = dx + ncol;
    int rc = 0;
    if (hipMemcpyAsync(dbk, bucket_host, nrow * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(dsg, sgn_host, nrow, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = 1;
    if (!rc) {
        hipLaunchKernelGGL(k_sketch, dim3((unsigned)((ncol + SK_COLS - 1) / SK_COLS)), dim3(256), 0, ctx->stream,
                           g->dev, nrow, ncol, g->ld, dbk, dsg, kp, dsk, dx, dx2);
        if (hipGetLastError() != hipSuccess) rc = 1;
    }
    if (!rc && sketch_host && hipMemcpyAsync(sketch_host, dsk, b_sk, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = 1;
    if (!rc && xsum_host && hipMemcpyAsync(xsum_host, dx, b_s, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = 1;
    if (!rc && x2sum_host && hipMemcpyAsync(x2sum_host, dx2, b_s, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) rc = 1;
    (void)hipStreamSynchronize(ctx->stream); (void)hipFree(d);
    if (rc) SFG_FAIL(ctx, "sfg_sketch: device operation failed");
    return 0;
}
