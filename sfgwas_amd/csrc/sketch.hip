// sketch.hip — the plaintext pass of randomized PCA over the local genotypes (gwas/pca.go:152-162):
//     localSketch[bucket[i]][j] += sgn[i] * x[i][j]   (fp64),   xsum[j] += uint64(x),   x2sum[j] += uint64(x*x)
// i.e. sketch = S (kp x n_ind, one +-1 per column) times X (n_ind x m_snp).  This is the one place on the path
// where the matrix cores are used: the count-sketch projection is an exact small-integer GEMM.  Both factors are int8
// (S is one-hot +-1, X the raw genotype bytes), so it runs on v_mfma_i32_16x16x64_i8 with exact int32 accumulation
// (flushed into fp64 accumulators every 2^16 rows: |partial| <= 2^30 for any int8 operands); the fp64 result the reference forms is the same
// integer.  Round 1 used v_mfma_f64_16x16x4_f64 (1.3 TB/s: bound by the fp64 matrix rate and the int8 -> fp64
// conversions); the int8 form leaves the kernel to the HBM stream.
// One 256-thread workgroup owns a 256-column strip of X over ALL rows (no atomics); genotype tiles of 64 rows x 256 B are
// staged through LDS with 16-byte loads.  A lane reads 16 row-dwords of its 4 columns, transposes them with v_perm_b32 into
// the four 16-byte B operands (k = 16 consecutive rows of one column), and takes the column moments from the same registers
// with v_dot4_i32_i8 - with Go's integer semantics (uint64(int8) sign-extends; x*x wraps in int8: the dot product of a dword
// with itself is used when it is < 128, which proves no square wrapped, else the four bytes are squared one by one).
#include "common.hpp"
#include "kernels.hpp"

typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int SK_ROWS = 64, SK_COLS = 256, SK_CHUNK = 16384;      // rows per workgroup: a multiple of SK_ROWS, below the int32 flush period

__global__ void __launch_bounds__(256) k_sketch(const int8_t *X, size_t nrow, size_t ncol, size_t ld, const int32_t *bucket, const int8_t *sgn,
                                                int kp, double *sketch, u64 *xsum, u64 *x2sum) {
    // tile[row][dword]: the dword index is XORed with ((row >> 4) & 3) << 4, so the four 16-row groups a wave reads together sit in four different bank quarters
    __shared__ __attribute__((aligned(16))) unsigned tile2[2][SK_ROWS][SK_COLS / 4];         // double-buffered: one barrier per 64 rows
    __shared__ __attribute__((aligned(16))) unsigned char bk2[2][SK_ROWS], sg2[2][SK_ROWS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, cd = lane & 15, kb = lane >> 4;
    const size_t col0 = (size_t)blockIdx.x * SK_COLS;
    v4i acc[4];                                  // column t of the lane's dword: 16 buckets x 16 dword-columns, rows (buckets) 4 kb + r
    double dacc[4][4];
#pragma unroll
    for (int t = 0; t < 4; t++) { acc[t] = (v4i){0, 0, 0, 0}; for (int r = 0; r < 4; r++) dacc[t][r] = 0.0; }
    long long s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    int p1[4] = {0, 0, 0, 0}, p2[4] = {0, 0, 0, 0};
    const unsigned b4 = (unsigned)cd * 0x01010101u;              // this lane's bucket (A-operand row) in every byte
    size_t since_flush = 0; bool wrapped = false;
    // rows [rbeg, rend) of this workgroup's chunk (grid.y): the chunks' integer partial results are combined with exact atomics
    const size_t rbeg = (size_t)blockIdx.y * SK_CHUNK, rend = rbeg + SK_CHUNK < nrow ? rbeg + SK_CHUNK : nrow;
    // the next tile's 16-byte pieces (4 per thread) and its bucket / sign bytes travel while the current tile is multiplied
    uint4 nx[4]; unsigned char nbk = 16, nsg = 0;
    auto fetch = [&](size_t r0) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int e = tid + 256 * u, rr = e / (SK_COLS / 16), cq = e % (SK_COLS / 16);
            const size_t row = r0 + rr, col = col0 + (size_t)cq * 16;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row < rend) {
                if (col + 16 <= ncol && (reinterpret_cast<uintptr_t>(X + row * ld + col) & 15) == 0) v = *reinterpret_cast<const uint4 *>(X + row * ld + col);
                else { int8_t b[16]; for (int k = 0; k < 16; k++) b[k] = col + k < ncol ? X[row * ld + col + k] : (int8_t)0; v = *reinterpret_cast<uint4 *>(b); }
            }
            nx[u] = v;
        }
        if (tid < SK_ROWS) { const size_t row = r0 + tid; nbk = row < rend ? (unsigned char)bucket[row] : (unsigned char)16; nsg = row < rend ? (unsigned char)sgn[row] : (unsigned char)0; }
    };
    fetch(rbeg);
    int buf = 0;
    for (size_t r0 = rbeg; r0 < rend; r0 += SK_ROWS, buf ^= 1) {
        unsigned (*tile)[SK_COLS / 4] = tile2[buf]; unsigned char *bk8 = bk2[buf], *sg8 = sg2[buf];
        // (the buffer written here was last read two iterations ago: the barrier of the previous iteration lies in between)
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int e = tid + 256 * u, rr = e / (SK_COLS / 16), cq = e % (SK_COLS / 16);
            *reinterpret_cast<uint4 *>(&tile[rr][(cq ^ (((rr >> 4) & 3) << 2)) * 4]) = nx[u];
        }
        if (tid < SK_ROWS) { bk8[tid] = nbk; sg8[tid] = nsg; }
        __syncthreads();
        if (r0 + SK_ROWS < rend) fetch(r0 + SK_ROWS);
        // A operand: S[bucket = cd][k = 16 kb + j] = sgn of row k if its bucket is cd, else 0 (bucket bytes are <= 16: no carry between bytes below)
        v4i A;
        {
            const uint4 bw = *reinterpret_cast<const uint4 *>(&bk8[16 * kb]), sw = *reinterpret_cast<const uint4 *>(&sg8[16 * kb]);
            const unsigned bwv[4] = {bw.x, bw.y, bw.z, bw.w}, swv[4] = {sw.x, sw.y, sw.z, sw.w};
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const unsigned x = bwv[g] ^ b4;
                const unsigned z = ~(x + 0x7F7F7F7Fu) & 0x80808080u;            // 0x80 where the byte of x is 0
                A[g] = (int)(swv[g] & ((z << 1) - (z >> 7)));
            }
        }
        // B operands: 16 row-dwords of the lane's 4 columns -> four columns x 16 rows
        unsigned W[16], Bt[4][4];
#pragma unroll
        for (int k = 0; k < 16; k++) W[k] = tile[16 * kb + k][(wave * 16 + cd) ^ (kb << 4)];
#pragma unroll
        for (int g = 0; g < 4; g++) {
            unsigned o[4]; bytes_tr4(W[4 * g], W[4 * g + 1], W[4 * g + 2], W[4 * g + 3], o);
#pragma unroll
            for (int t = 0; t < 4; t++) Bt[t][g] = o[t];
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const v4i B = {(int)Bt[t][0], (int)Bt[t][1], (int)Bt[t][2], (int)Bt[t][3]};
            acc[t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(A, B, acc[t], 0, 0, 0);
            // column moments of column 4 cd + t over these 16 rows
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int w = (int)Bt[t][g];
                p1[t] = __builtin_amdgcn_sdot4(w, 0x01010101, p1[t], false);           // += the four signed bytes   (uint64(row[j]), pca.go:158)
                const int sq = __builtin_amdgcn_sdot4(w, w, 0, false);                  // sum of the four squares: exact when < 128 (then none wrapped in int8)
                p2[t] += sq; wrapped |= sq >= 128;
            }
        }
        if (wrapped) {                                           // rare (never for dosages 0..2): redo the dwords whose squares may have wrapped, byte by byte (:159)
            wrapped = false;
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int g = 0; g < 4; g++) {
                    const int w = (int)Bt[t][g];
                    const int sq = __builtin_amdgcn_sdot4(w, w, 0, false);
                    if (sq >= 128) p2[t] += sq_sum4_i8(w) - sq;
                }
        }
        since_flush += SK_ROWS;
        if (since_flush >= (1u << 16)) {                         // int32 partial sums stay below 2^31 for any int8 signs (128 * 128 * 2^16 = 2^30)
            since_flush = 0;
#pragma unroll
            for (int t = 0; t < 4; t++) {
#pragma unroll
                for (int r = 0; r < 4; r++) { dacc[t][r] += (double)acc[t][r]; acc[t][r] = 0; }
                s1[t] += p1[t]; s2[t] += p2[t]; p1[t] = 0; p2[t] = 0;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; t++) {
        s1[t] += p1[t]; s2[t] += p2[t];
        // the four 16-row groups (kb) of a column sit in lanes cd, cd + 16, cd + 32, cd + 48
        s1[t] += __shfl_xor(s1[t], 16); s1[t] += __shfl_xor(s1[t], 32);
        s2[t] += __shfl_xor(s2[t], 16); s2[t] += __shfl_xor(s2[t], 32);
        const size_t col = col0 + (size_t)wave * 64 + 4 * cd + t;
        if (kb == 0 && col < ncol) { atomicAdd((unsigned long long *)&xsum[col], (unsigned long long)s1[t]); atomicAdd((unsigned long long *)&x2sum[col], (unsigned long long)s2[t]); }
        // C/D layout of the 16x16 integer MFMA: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int row = 4 * kb + r;
            if (row < kp && col < ncol) atomicAdd(&sketch[(size_t)row * ncol + col], dacc[t][r] + (double)acc[t][r]);       // integer-valued: exact in any order
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
    const size_t b_bk = (nrow * 4 + 15) & ~(size_t)15, b_sg = (nrow + 15) & ~(size_t)15,            // (the 64-bit atomics on the result arrays need them 8-byte aligned)
                  b_sk = (size_t)kp * ncol * 8, b_s = ncol * 8;
    SFG_HIP(ctx, hipMalloc(&d, b_bk + b_sg + b_sk + 2 * b_s));
    int32_t *dbk = (int32_t *)d; int8_t *dsg = (int8_t *)(d + b_bk); double *dsk = (double *)(d + b_bk + b_sg);
    u64 *dx = (u64 *)(d + b_bk + b_sg + b_sk), *dx2 = dx + ncol;
    int rc = 0;
    if (hipMemcpyAsync(dbk, bucket_host, nrow * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(dsg, sgn_host, nrow, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = 1;
    if (!rc && hipMemsetAsync(dsk, 0, b_sk + 2 * b_s, ctx->stream) != hipSuccess) rc = 1;
    if (!rc) {
        hipLaunchKernelGGL(k_sketch, dim3((unsigned)((ncol + SK_COLS - 1) / SK_COLS), (unsigned)((nrow + SK_CHUNK - 1) / SK_CHUNK)), dim3(256), 0, ctx->stream,
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
