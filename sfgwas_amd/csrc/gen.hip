// gen.hip — synthetic data generators used by bench.py and tests (SURVEY.md §8d "Synthetic inputs").
// Everything is counter-mode splitmix64, so the same values can be regenerated on the host
// (the test-side generators use the same formulas) without moving data.
#include "common.hpp"
#include "kernels.hpp"

// ciphertext rows: element (row, x) = mulhi(splitmix(seed + row*N + x), q_row): a CKKS ciphertext is
// computationally uniform mod q.  rows are [nct][2][nl] with modulus index = row % nl.
__global__ void __launch_bounds__(256) k_fill_uniform(u64 *rows, int nl, u64 seed, const ModConst *modc, int per_ct_seed) {
    const int N = SFG_N; const size_t row = blockIdx.x / (N / 256);
    const int x = (int)(blockIdx.x % (N / 256)) * 256 + threadIdx.x;
    const int m = (int)(row % nl);
    // per_ct_seed: ciphertext j uses seed + j and indexes elements inside the ciphertext, which is what
    // orc_fill_uniform(ring, level, seed + j) produces on the host
    u64 sd = seed, idx = row * N + x;
    if (per_ct_seed) { const size_t ct = row / (2 * (size_t)nl); sd = seed + ct; idx = (row % (2 * (size_t)nl)) * N + x; }
    rows[row * N + x] = __umul64hi(splitmix_at(sd, idx), modc[m].qi);
}
extern "C" int sfg_fill_uniform_ct_dev(sfg_ctx *ctx, uint64_t *ct, int nct, int level, uint64_t seed) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (level < 0 || level >= ctx->nq || nct < 0) SFG_FAIL(ctx, "fill_uniform_ct: level %d / count %d out of range", level, nct);
    const int nl = level + 1; const size_t rows = (size_t)nct * 2 * nl;
    if (!rows) return 0;
    hipLaunchKernelGGL(k_fill_uniform, dim3((unsigned)(rows * (SFG_N / 256))), dim3(256), 0, ctx->stream, (u64 *)ct, nl, (u64)seed, ctx->modc, 1);
    SFG_HIP(ctx, hipGetLastError());
    return 0;
}

// genotypes: X[i][j] ~ Binomial(2, p_j), p_j ~ U(0.05, 0.5), missing (-1) with probability 1/128.
// p_j and the three uniforms are 24-bit fractions of splitmix words, so the host can reproduce them exactly.
__device__ __host__ static inline unsigned geno_pj24(u64 h) {            // p_j * 2^24, p_j in [0.05, 0.5)
    const unsigned lo = 838861u, span = 7549747u;                         // 0.05 * 2^24, 0.45 * 2^24
    return lo + (unsigned)(((h >> 40) * (u64)span) >> 24);
}
// element (i, jg) of the GLOBAL nrow x ncol_global matrix, written for the column window [col0, col0 + ncol) at g[i*ld + (jg - col0)]:
// a rank that owns a SNP-column window of X holds exactly the bytes the single-GPU run holds there
__global__ void __launch_bounds__(256) k_fill_geno(int8_t *g, size_t row0, size_t ncol, size_t ld, size_t col0, size_t ncol_global, u64 seed) {
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x, i = row0 + blockIdx.y;
    if (j >= ncol) return;
    const size_t jg = col0 + j;
    const unsigned pj = geno_pj24(splitmix_at(seed ^ 0xC01C01ULL, jg));
    const u64 h = splitmix_at(seed, i * ncol_global + jg);
    const unsigned u0 = (unsigned)(h & 0xFFFFFF), u1 = (unsigned)((h >> 24) & 0xFFFFFF), um = (unsigned)(h >> 57);
    int8_t v = (int8_t)((u0 < pj) + (u1 < pj));
    if (um == 0) v = -1;
    g[i * ld + j] = v;
}
extern "C" int sfg_fill_geno_window_dev(sfg_ctx *ctx, int8_t *geno, size_t nrow, size_t ncol, size_t ld, size_t col0, size_t ncol_global, uint64_t seed) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (!nrow || !ncol) return 0;
    if (nrow > 2147483647ULL) SFG_FAIL(ctx, "sfg_fill_geno: too many rows");
    if (ld < ncol || col0 + ncol > ncol_global) SFG_FAIL(ctx, "sfg_fill_geno: bad column window");
    for (size_t r0 = 0; r0 < nrow; r0 += 65535) {          // grid.y is limited to 65535: row bands
        const size_t nr = nrow - r0 < 65535 ? nrow - r0 : 65535;
        hipLaunchKernelGGL(k_fill_geno, dim3((unsigned)((ncol + 255) / 256), (unsigned)nr), dim3(256), 0, ctx->stream, geno, r0, ncol, ld, col0, ncol_global, (u64)seed);
        SFG_HIP(ctx, hipGetLastError());
    }
    return 0;
}
extern "C" int sfg_fill_geno_dev(sfg_ctx *ctx, int8_t *geno, size_t nrow, size_t ncol, uint64_t seed) {
    return sfg_fill_geno_window_dev(ctx, geno, nrow, ncol, ncol, 0, ncol, seed);
}

// rotation keys with uniform random words (timing-equivalent to real keys)
__global__ void __launch_bounds__(256) k_fill_key(u64 *key, int nmod, u64 seed, const ModConst *modc) {
    const int N = SFG_N; const size_t row = blockIdx.x / (N / 256);
    const int x = (int)(blockIdx.x % (N / 256)) * 256 + threadIdx.x;
    const int m = (int)(row % nmod);
    key[row * N + x] = __umul64hi(splitmix_at(seed, row * N + x), modc[m].qi);
}
extern "C" int sfg_fill_rotkeys_synthetic(sfg_ctx *ctx, const int *rot_left, int nrot, uint64_t seed) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    const int N = SFG_N; const size_t rows = (size_t)ctx->beta * 2 * ctx->nmod, words = rows * N;
    for (int k = 0; k < nrot; k++) {
        const u64 g = sfg_galois_for_rotation(ctx, rot_left[k]);
        if (ctx->rotkeys().count(g)) continue;
        RotKey rk;
        SFG_HIP(ctx, hipMalloc(&rk.key_dev, words * 8));
        SFG_HIP(ctx, hipMalloc(&rk.index_dev, N * sizeof(uint16_t)));
        hipLaunchKernelGGL(k_fill_key, dim3((unsigned)(rows * (N / 256))), dim3(256), 0, ctx->stream, rk.key_dev, ctx->nmod, (u64)seed + g, ctx->modc);
        SFG_HIP(ctx, hipGetLastError());
        std::vector<uint16_t> idx(N); const u64 mask = 2ULL * N - 1;
        for (int i = 0; i < N; i++) { u64 t1 = 2ULL * h_brev((uint32_t)i, SFG_LOGN) + 1; u64 t2 = ((g * t1 & mask) - 1) >> 1; idx[i] = (uint16_t)h_brev((uint32_t)t2, SFG_LOGN); }
        SFG_HIP(ctx, hipMemcpyAsync(rk.index_dev, idx.data(), N * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->stream));
        SFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        ctx->rotkeys()[g] = rk;
    }
    return 0;
}
