// matmul.hip — orchestration of the encrypted-row-vector x int8 genotype matrix product
// (MatMult4Stream gwas/matmult.go:1238-1505, MatMult4StreamCompute :1043-1236, MatMult4StreamPreprocess :914-1041).
//
// The reference streams one diagonal at a time through s*d*m_ct u128 accumulator polynomials.  On the GPU the
// same canonical sums are produced block by block:
//   for every block row bi of the operand:   rotCache[bi] = { RotateRight(A[i][bi], -baby) }      (rotate.hip)
//   for every 8192x8192 block (bi, j):        P = encode(all 8192 diagonals of the block)           (encode.hip, ntt.hip)
//                                             acc[j][giant][i] += sum_baby rotCache[bi][baby][i] * P[giant*d + baby]   (mac.hip)
//   finalize:  out[i][j] = sum_giant RotateRight(acc[j][giant][i], -giant*d)                       (rotate.hip)
// acc holds canonical residues (8 B per coefficient instead of the reference's 16 B lazy u128), and only for the
// block columns of the current group, so 100k x 1M fits one GPU's HBM.  The genotype matrix is kept once in HBM
// as int8 and serves both X and X^T (SFG_TRANSPOSE).  Active baby/giant tables follow matmult.go:1329-1336.
#include "common.hpp"
#include "kernels.hpp"
#include "i8_move.hpp"            // PtRide
#include <algorithm>
#include <string>

static inline int ceil_div(size_t a, size_t b) { return (int)((a + b - 1) / b); }
static inline int diag_bool(int r, int c, int dim, int index) {          // GetDiagBool, matmult.go:627-631
    index %= dim; if (index < 0) index += dim;
    return (dim + 1 - r) <= index || index <= c - 1;
}

struct Shape {
    size_t nrow, ncol;      // logical operand dims (after optional transpose)
    int nbr, m_ct;
    bool transposed;
    const int8_t *dev; size_t ld;
    const sfg_geno *g;      // the resident matrix (2-bit packed matrices are expanded block by block, see matmul_accumulate)
    int rows_of(int bi) const { return (int)(std::min((size_t)(bi + 1) * SFG_SLOTS, nrow) - (size_t)bi * SFG_SLOTS); }
    int cols_of(int bj) const { return (int)(std::min((size_t)(bj + 1) * SFG_SLOTS, ncol) - (size_t)bj * SFG_SLOTS); }
    // pointer to the stored top-left element of logical block (bi, bj)
    const int8_t *block(int bi, int bj) const {
        return transposed ? dev + (size_t)bj * SFG_SLOTS * ld + (size_t)bi * SFG_SLOTS
                          : dev + (size_t)bi * SFG_SLOTS * ld + (size_t)bj * SFG_SLOTS;
    }
};
static Shape make_shape(const sfg_geno *g, unsigned flags) {
    Shape sh; sh.transposed = (flags & SFG_TRANSPOSE) != 0;
    sh.nrow = sh.transposed ? g->ncol : g->nrow; sh.ncol = sh.transposed ? g->nrow : g->ncol;
    sh.nbr = ceil_div(sh.nrow, SFG_SLOTS); sh.m_ct = ceil_div(sh.ncol, SFG_SLOTS);
    sh.dev = g->dev; sh.ld = g->ld; sh.g = g; return sh;
}

// ---------------------------------------------------------------- small kernels
// DropLevel (crypto/basics.go:806-824 at matmult.go:1055,1258): keep the first nl rows of each polynomial
__global__ void __launch_bounds__(256) k_drop_level(const u64 *in, u64 *out, int nl_in, int nl) {
    const int N = SFG_N; const size_t row = blockIdx.x / (N / 256);           // output row over [ct][2][nl]
    const size_t ctp = row / nl, m = row % nl;
    const size_t x = (blockIdx.x % (N / 256)) * 256 + threadIdx.x;
    out[row * N + x] = in[(ctp * nl_in + m) * N + x];
}
// out[i][j][p][l][x] (+)= sum_{k < ngiant} rot[(k*s + i)][p][l][x]   (rot holds only the aligned giants, compactly)
__global__ void __launch_bounds__(256) k_sum_giants(const u64 *rot, int ngiant, int s, int L, u64 *out, size_t out_i_stride,
                                                    int accumulate, const ModConst *modc) {
    const int N = SFG_N; const size_t row = blockIdx.x / (N / 256);           // row over [i][2][L]
    const int i = (int)(row / (2 * L)), pl = (int)(row % (2 * L)), l = pl % L;
    const size_t x = (blockIdx.x % (N / 256)) * 256 + threadIdx.x;
    const double q = modc[l].q, qinv = modc[l].qinv;
    double acc = accumulate ? u64_to_f64(out[(size_t)i * out_i_stride + (size_t)pl * N + x]) : 0.0;
    for (int g = 0; g < ngiant; g++) {
        acc += u64_to_f64(rot[(((size_t)g * s + i) * 2 * L + pl) * N + x]);
        if ((g & 31) == 31) acc = pred(acc, q, qinv);
    }
    out[(size_t)i * out_i_stride + (size_t)pl * N + x] = f64_to_u64(canon(acc, q, qinv));
}
__global__ void __launch_bounds__(256) k_reduce_rows(u64 *rows, int L, const ModConst *modc) {
    const int N = SFG_N; const size_t row = blockIdx.x / (N / 256); const int l = (int)(row % L);
    const size_t off = row * N + (blockIdx.x % (N / 256)) * 256 + threadIdx.x;
    rows[off] = rows[off] % modc[l].qi;
}
// zero plaintexts for the panel slots the encoder does not write: nseg segments of seg_words words, pitch_words apart; a word of
// modulus row l (= (index / prow) % L inside a segment, segments start on a plaintext boundary) is PACKED_ZERO when packed_mask has bit l
__global__ void __launch_bounds__(256) k_pt_zero(u64 *pt, size_t pitch_words, size_t seg_words, int L, size_t prow, unsigned packed_mask) {
    const size_t seg = blockIdx.y, x = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (x >= seg_words) return;
    const int l = (int)((x / prow) % L);
    pt[seg * pitch_words + x] = ((packed_mask >> l) & 1u) ? PACKED_ZERO : 0ULL;
}
// the same for a K-major panel ([column][plane][128-byte coefficient block][k < K][128 B], kernels.hpp PT_KMAJOR): rows [k0, k1) of columns [n0, n1), every plane
__global__ void __launch_bounds__(256) k_pt_zero_kmajor(uint8_t *pt, int K, int planes, int n0, int k0, int nk) {
    const size_t blk = ((size_t)(n0 + blockIdx.y) * planes * 64 + blockIdx.x) * (size_t)K * 128 + (size_t)k0 * 128;
    uint4 *p = reinterpret_cast<uint4 *>(pt + blk);
    for (int i = threadIdx.x; i < nk * 8; i += 256) p[i] = make_uint4(0, 0, 0, 0);
}
static int launch_pt_zero_kmajor(sfg_ctx *ctx, u64 *pt, int K, int planes, int n0, int n1, int k0, int k1) {
    if (n1 <= n0 || k1 <= k0) return 0;
    hipLaunchKernelGGL(k_pt_zero_kmajor, dim3((unsigned)(planes * 64), (unsigned)(n1 - n0)), dim3(256), 0, ctx->stream, reinterpret_cast<uint8_t *>(pt), K, planes, n0, k0, k1 - k0);
    SFG_HIP(ctx, hipGetLastError());
    return 0;
}
static int launch_pt_zero(sfg_ctx *ctx, u64 *pt, size_t pitch_words, size_t seg_words, int nseg, int L, size_t prow, unsigned packed_mask) {
    if (!packed_mask) {
        if (nseg == 1) SFG_HIP(ctx, hipMemsetAsync(pt, 0, seg_words * 8, ctx->stream));
        else SFG_HIP(ctx, hipMemset2DAsync(pt, pitch_words * 8, 0, seg_words * 8, nseg, ctx->stream));
        return 0;
    }
    hipLaunchKernelGGL(k_pt_zero, dim3((unsigned)((seg_words + 255) / 256), (unsigned)nseg), dim3(256), 0, ctx->stream, pt, pitch_words, seg_words, L, prow, packed_mask);
    SFG_HIP(ctx, hipGetLastError());
    return 0;
}
// P2: per-column sum and sum of squares after missing -> 0 (matmult.go:1292-1300).  A workgroup takes a 256-column strip of a chunk of rows:
// thread = (16-byte column group tid & 15, row lane tid >> 4), so a wave's load is four rows of 256 contiguous bytes.  Four rows at a time are
// byte-transposed in registers (v_perm_b32), so that a dword holds four rows of ONE column: negatives are cleared four at a time, the sum is one
// v_dot4_i32_i8, the squares (Go: float64(int8(x*x)), wraps for x >= 12) the dot product of the dword with itself when that proves nothing wrapped.
// Partial sums are integers: the fp64 atomics that combine the row chunks are exact.
constexpr int CS_ROWS = 4096;                     // rows per workgroup (256 per thread: int32 partial sums)
__global__ void __launch_bounds__(256) k_colsums(const int8_t *g, size_t nrow, size_t ncol, size_t ld, double *sum, double *sqsum) {
    __shared__ int red[2][16][256 + 1];
    const int tid = threadIdx.x, cg = tid & 15, rl = tid >> 4;
    const size_t col = (size_t)blockIdx.x * 256 + (size_t)cg * 16, r0 = (size_t)blockIdx.y * CS_ROWS, r1 = r0 + CS_ROWS < nrow ? r0 + CS_ROWS : nrow;
    int p1[16], p2[16];
#pragma unroll
    for (int k = 0; k < 16; k++) p1[k] = p2[k] = 0;
    const bool vec = col + 16 <= ncol && ((reinterpret_cast<uintptr_t>(g) | ld) & 15) == 0;
    auto load = [&](size_t i) -> uint4 {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (i >= r1) return v;
        if (vec) v = *reinterpret_cast<const uint4 *>(g + i * ld + col);
        else if (col < ncol) { int8_t b[16]; for (int k = 0; k < 16; k++) b[k] = col + k < ncol ? g[i * ld + col + k] : (int8_t)0; v = *reinterpret_cast<uint4 *>(b); }
        return v;
    };
    for (size_t i = r0 + rl; i < r1; i += 64) {               // rows i, i + 16, i + 32, i + 48
        const uint4 v0 = load(i), v1 = load(i + 16), v2 = load(i + 32), v3 = load(i + 48);
        const unsigned a0[4] = {v0.x, v0.y, v0.z, v0.w}, a1[4] = {v1.x, v1.y, v1.z, v1.w}, a2[4] = {v2.x, v2.y, v2.z, v2.w}, a3[4] = {v3.x, v3.y, v3.z, v3.w};
#pragma unroll
        for (int q = 0; q < 4; q++) {
            unsigned o[4]; bytes_tr4(a0[q], a1[q], a2[q], a3[q], o);                     // o[t]: column 4 q + t, four rows
#pragma unroll
            for (int t = 0; t < 4; t++) {
                unsigned w = o[t];
                const unsigned m = w & 0x80808080u; w &= ~((m << 1) - (m >> 7));           // missing (negative) -> 0
                p1[4 * q + t] = __builtin_amdgcn_sdot4((int)w, 0x01010101, p1[4 * q + t], false);
                p2[4 * q + t] += sq_sum4_i8((int)w);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 16; k++) { red[0][rl][cg * 16 + k] = p1[k]; red[1][rl][cg * 16 + k] = p2[k]; }
    __syncthreads();
    long long s1 = 0, s2 = 0;
#pragma unroll
    for (int r = 0; r < 16; r++) { s1 += red[0][r][tid]; s2 += red[1][r][tid]; }
    const size_t j = (size_t)blockIdx.x * 256 + tid;
    if (j < ncol) {
        if (sum) atomicAdd(&sum[j], (double)s1);
        if (sqsum) atomicAdd(&sqsum[j], (double)s2);
    }
}

// C6: crypto.DropLevel / eval.DropLevelNew (basics.go:806-824): keep the first level_out+1 moduli rows of each polynomial.
// level_out == level_in is the CopyNew branch; level_out > level_in fails like the reference (log.Fatalf).
extern "C" int sfg_ct_drop_level_dev(sfg_ctx *ctx, const uint64_t *in, uint64_t *out, int nct, int level_in, int level_out) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (level_out > level_in) SFG_FAIL(ctx, "DropLevel: requested level %d when input is %d", level_out, level_in);
    if (level_out < 0 || level_in >= ctx->nq || nct < 0) SFG_FAIL(ctx, "DropLevel: level out of range");
    if (!nct) return 0;
    const int nl = level_out + 1, nl_in = level_in + 1;
    if (nl == nl_in) { SFG_HIP(ctx, hipMemcpyAsync(out, in, (size_t)nct * 2 * nl * SFG_N * 8, hipMemcpyDeviceToDevice, ctx->stream)); return 0; }
    hipLaunchKernelGGL(k_drop_level, dim3((unsigned)((size_t)nct * 2 * nl * (SFG_N / 256))), dim3(256), 0, ctx->stream, (const u64 *)in, (u64 *)out, nl_in, nl);
    SFG_HIP(ctx, hipGetLastError());
    return 0;
}
extern "C" int sfg_memcpy_d2d(sfg_ctx *ctx, void *dst, const void *src, size_t bytes) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    SFG_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}

// ---------------------------------------------------------------- genotype residency
extern "C" int sfg_geno_upload(sfg_ctx *ctx, const int8_t *host, size_t nrow, size_t ncol, size_t ld, sfg_geno **out) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (!nrow || !ncol || ld < ncol) SFG_FAIL(ctx, "sfg_geno_upload: bad dimensions");
    int8_t *d = nullptr;
    SFG_HIP(ctx, hipMalloc(&d, nrow * ncol));
    SFG_HIP(ctx, hipMemcpy2DAsync(d, ncol, host, ld, ncol, nrow, hipMemcpyHostToDevice, ctx->stream));
    SFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    sfg_geno *g = new sfg_geno(); g->dev = d; g->nrow = nrow; g->ncol = ncol; g->ld = ncol; g->owned = true;
    *out = g; return 0;
}
// ---- row-streamed registration.  The reference never holds its matrix: MatMult4StreamPreprocess (matmult.go:914-1041) pulls one row at a time out of
// GenoFileStream.NextRow (filestream.go:414-426).  sfg_geno_create makes the resident matrix, sfg_geno_write_rows fills it chunk by chunk from whatever staging
// buffer the host keeps (sfg_pinned_alloc gives a page-locked one), so no host allocation scales with nrow * ncol.  sfg_geno_compare_rows answers "is the matrix
// now arriving row by row the one already resident - or its transpose?" (pca.go:112-113 registers X, then X^T from a second file) by comparing each chunk with the
// resident copy ON the device, entry by entry: X^T is then recognised without being held anywhere, and the one int8 copy serves both orientations.
extern "C" int sfg_geno_create(sfg_ctx *ctx, size_t nrow, size_t ncol, sfg_geno **out) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (!out) SFG_FAIL(ctx, "sfg_geno_create: null result pointer");
    *out = nullptr;
    if (!nrow || !ncol) SFG_FAIL(ctx, "sfg_geno_create: bad dimensions");
    int8_t *d = nullptr;
    SFG_TRY(sfg_malloc(ctx, (void **)&d, nrow * ncol));
    sfg_geno *g = new sfg_geno(); g->dev = d; g->nrow = nrow; g->ncol = ncol; g->ld = ncol; g->owned = true;
    *out = g; return 0;
}
extern "C" int sfg_geno_write_rows(sfg_ctx *ctx, sfg_geno *g, size_t row0, size_t nrows, const int8_t *rows_host, size_t ld) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (!g || !g->owned || g->packed) SFG_FAIL(ctx, "sfg_geno_write_rows: not a matrix made by sfg_geno_create");
    if (!nrows) return 0;
    if (!rows_host || ld < g->ncol || row0 > g->nrow || nrows > g->nrow - row0) SFG_FAIL(ctx, "sfg_geno_write_rows: rows [%zu, %zu) of a %zu x %zu matrix, row stride %zu", row0, row0 + nrows, g->nrow, g->ncol, ld);
    if (!g->ptc.empty()) SFG_FAIL(ctx, "sfg_geno_write_rows: the matrix has cached plaintexts (sfg_geno_set_plaintext_cache): it must not change");
    SFG_HIP(ctx, hipMemcpy2DAsync(const_cast<int8_t *>(g->dev) + row0 * g->ld, g->ld, rows_host, ld, g->ncol, nrows, hipMemcpyHostToDevice, ctx->stream));
    SFG_HIP(ctx, hipStreamSynchronize(ctx->stream));           // the caller refills its staging buffer next
    return 0;
}
// chunk[r][c] (dense, ncol_l columns) against logical row row0 + r of the stored matrix (transposed: stored[c][row0 + r]); consecutive lanes walk the STORED rows
__global__ void __launch_bounds__(256) k_geno_compare(const int8_t *chunk, const int8_t *stored, size_t ld, size_t row0, size_t nrows, size_t ncol_l, int transposed, unsigned long long *ndiff) {
    const size_t total = nrows * ncol_l;
    unsigned long long bad = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        size_t r, c;
        if (transposed) { r = i % nrows; c = i / nrows; } else { r = i / ncol_l; c = i % ncol_l; }
        const int8_t want = transposed ? stored[c * ld + row0 + r] : stored[(row0 + r) * ld + c];
        bad += chunk[r * ncol_l + c] != want;
    }
    if (bad) atomicAdd(ndiff, bad);
}
extern "C" int sfg_geno_compare_rows(sfg_ctx *ctx, const sfg_geno *g, unsigned flags, size_t row0, size_t nrows, const int8_t *rows_host, size_t ld, uint64_t *ndiff) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (!g || g->packed || !ndiff) SFG_FAIL(ctx, "sfg_geno_compare_rows: needs an unpacked resident matrix and a result pointer");
    const bool tr = flags & SFG_TRANSPOSE;
    const size_t nrow_l = tr ? g->ncol : g->nrow, ncol_l = tr ? g->nrow : g->ncol;
    if (!nrows) return 0;
    if (!rows_host || ld < ncol_l || row0 > nrow_l || nrows > nrow_l - row0) SFG_FAIL(ctx, "sfg_geno_compare_rows: rows [%zu, %zu) of a %zu x %zu matrix, row stride %zu", row0, row0 + nrows, nrow_l, ncol_l, ld);
    int8_t *chunk = nullptr; unsigned long long *cnt = nullptr;
    SFG_TRY(sfg_scratch(ctx, "geno.cmp", nrows * ncol_l, (void **)&chunk));
    SFG_TRY(sfg_scratch(ctx, "geno.cmpcnt", 8, (void **)&cnt));
    SFG_HIP(ctx, hipMemsetAsync(cnt, 0, 8, ctx->stream));
    SFG_HIP(ctx, hipMemcpy2DAsync(chunk, ncol_l, rows_host, ld, ncol_l, nrows, hipMemcpyHostToDevice, ctx->stream));
    const size_t total = nrows * ncol_l;
    hipLaunchKernelGGL(k_geno_compare, dim3((unsigned)std::min<size_t>((total + 255) / 256, 65536)), dim3(256), 0, ctx->stream, (const int8_t *)chunk, g->dev, g->ld, row0, nrows, ncol_l, tr ? 1 : 0, cnt);
    SFG_HIP(ctx, hipGetLastError());
    unsigned long long h = 0;
    SFG_HIP(ctx, hipMemcpyAsync(&h, cnt, 8, hipMemcpyDeviceToHost, ctx->stream));
    SFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    *ndiff += h;
    return 0;
}
// page-locked host memory for a caller's staging buffer (the Go shim fills it from GenoFileStream.NextRow): H2D copies from it run at the PCIe rate
extern "C" int sfg_pinned_alloc(sfg_ctx *ctx, void **host_ptr, size_t bytes) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (!host_ptr || !bytes) SFG_FAIL(ctx, "sfg_pinned_alloc: bad arguments");
    SFG_HIP(ctx, hipHostMalloc(host_ptr, bytes, hipHostMallocDefault));
    return 0;
}
extern "C" int sfg_pinned_free(sfg_ctx *ctx, void *host_ptr) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (host_ptr) SFG_HIP(ctx, hipHostFree(host_ptr));
    return 0;
}
extern "C" int sfg_geno_from_device(sfg_ctx *ctx, const int8_t *dev, size_t nrow, size_t ncol, size_t ld, sfg_geno **out) {
    if (!nrow || !ncol || ld < ncol) SFG_FAIL(ctx, "sfg_geno_from_device: bad dimensions");
    sfg_geno *g = new sfg_geno(); g->dev = dev; g->nrow = nrow; g->ncol = ncol; g->ld = ld; g->owned = false;
    *out = g; return 0;
}
static void ptc_drop(sfg_ctx *ctx, const sfg_geno *g) {
    sfg_ctx *own = g->ptc_owner ? (sfg_ctx *)g->ptc_owner : ctx;
    own->ptc_genos.erase(g);
    if (g->ptc.empty() && !g->ptc_perm) return;
    (void)hipSetDevice(own->device); (void)sfg_sync_all(own);          // every queue of the owner (fills and hits may still be running on its encode queue)
    if (own != ctx) (void)sfg_sync_all(ctx);
    g->ptc.clear(); g->ptc_used = 0;
    if (g->ptc_arena) { (void)hipFree(g->ptc_arena); g->ptc_arena = nullptr; }
    if (g->ptc_perm) { (void)hipFree(g->ptc_perm); g->ptc_perm = nullptr; }
}
// An unprovable encoder rounding (2^-50 counter) was reported or reset: rows cached while it was outstanding cannot be told from the others, and the recovery
// ("re-derive, then reset") must not keep serving them - every cache this context owns forgets its rows (the arenas stay; later products refill them).
void sfg_ptc_invalidate_all(sfg_ctx *ctx) {
    for (const sfg_geno *g : ctx->ptc_genos) { g->ptc.clear(); g->ptc_used = 0; }
}
// The owning context goes away (sfg_ctx_destroy, after its queues have drained): every cache it owns is released and its matrices are detached, so that a later
// sfg_geno_free / sfg_geno_set_plaintext_cache on another context finds no dangling owner.  The matrices themselves stay valid (cache off).
void sfg_ptc_detach_all(sfg_ctx *ctx) {
    for (const sfg_geno *g : ctx->ptc_genos) {
        g->ptc.clear(); g->ptc_used = 0; g->ptc_budget = 0;
        if (g->ptc_arena) { (void)hipFree(g->ptc_arena); g->ptc_arena = nullptr; }
        if (g->ptc_perm) { (void)hipFree(g->ptc_perm); g->ptc_perm = nullptr; }
        g->ptc_owner = nullptr;
    }
    ctx->ptc_genos.clear();
}
extern "C" void sfg_geno_free(sfg_ctx *ctx, sfg_geno *g) {
    if (!g) return;
    ptc_drop(ctx, g);
    if (g->owned) { (void)hipSetDevice(ctx->device); (void)sfg_sync_all(ctx); (void)hipFree((void *)g->dev); }
    delete g;
}
// Plaintext coefficient cache of a resident matrix: up to max_bytes of HBM (512 MB per 8192 x 8192 block and SFG_SQUARE flavour), filled by the products that
// run after this call, hit by every later product over the same block - Q.X of the next power iteration, and Q'.X^T of this one: the diagonals of the transposed
// block are rotations of the cached ones (D'_t = rotL_t(D_{n-t})), a rotation of the slots is an automorphism of the plaintext, and the encoder's rounding
// commutes with it, so the NTT reads the cached row through a signed index permutation and produces the words a fresh encode would.  max_bytes = 0 drops it.
// The caller promises the matrix does not change while cached, and that `ctx` (its current stream) is the only context multiplying with this handle.
extern "C" int sfg_geno_set_plaintext_cache(sfg_ctx *ctx, const sfg_geno *g, size_t max_bytes) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (!g) SFG_FAIL(ctx, "sfg_geno_set_plaintext_cache: null matrix");
    if (!max_bytes) { ptc_drop(ctx, g); g->ptc_budget = 0; g->ptc_owner = nullptr; return 0; }
    if (g->ptc_owner && g->ptc_owner != (const void *)ctx) SFG_FAIL(ctx, "sfg_geno_set_plaintext_cache: the cache of this matrix belongs to another context (drop it there first)");
    if (!g->ptc_perm) {
        const int n = SFG_SLOTS, d = SFG_D; const unsigned M = 2u * SFG_N;
        std::vector<uint32_t> tab(n);
        std::vector<unsigned> pow5(n); pow5[0] = 1; for (int k = 1; k < n; k++) pow5[k] = (unsigned)(((u64)pow5[k - 1] * 5u) % M);     // 5 has order N/2 = n mod 2N
        for (int t = 0; t < n; t++) {
            const int u = (n - t) % n, k = ((t % d) + d * (u / d)) % n;
            tab[t] = (uint32_t)u | (uint32_t)pow5[k] << 16;                       // g = 5^k mod 2N  (< 2^15)
        }
        uint32_t *dv = nullptr;
        SFG_HIP(ctx, hipMalloc((void **)&dv, (size_t)n * 4));
        if (hipMemcpy(dv, tab.data(), (size_t)n * 4, hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(dv); SFG_FAIL(ctx, "sfg_geno_set_plaintext_cache: table upload failed"); }
        g->ptc_perm = dv;
    }
    const size_t slot_bytes = (size_t)SFG_SLOTS * SFG_SLOTS * 8, want = max_bytes / slot_bytes * slot_bytes;
    if (want != g->ptc_budget || !g->ptc_arena) {            // (re)sized: what was cached goes
        (void)hipStreamSynchronize(ctx->stream);
        g->ptc.clear(); g->ptc_used = 0;
        if (g->ptc_arena) { (void)hipFree(g->ptc_arena); g->ptc_arena = nullptr; }
        g->ptc_budget = 0;
        if (!want) SFG_FAIL(ctx, "sfg_geno_set_plaintext_cache: max_bytes is less than one block (%zu bytes)", slot_bytes);
        void *a = nullptr;
        if (hipMalloc(&a, want) != hipSuccess) { (void)hipGetLastError(); SFG_FAIL(ctx, "sfg_geno_set_plaintext_cache: cannot allocate %zu bytes", want); }
        g->ptc_arena = (double *)a; g->ptc_budget = want;
    }
    g->ptc_owner = ctx; ctx->ptc_genos.insert(g);
    return 0;
}
// blocks cached / bytes held / hits and fills since the cache was enabled
extern "C" int sfg_geno_plaintext_cache_stats(sfg_ctx *ctx, const sfg_geno *g, size_t *blocks, size_t *bytes, size_t *hits, size_t *fills) {
    if (!g) SFG_FAIL(ctx, "sfg_geno_plaintext_cache_stats: null matrix");
    if (blocks) *blocks = g->ptc.size();
    if (bytes) *bytes = g->ptc_used;
    if (hits) *hits = g->ptc_hits;
    if (fills) *fills = g->ptc_fills;
    return 0;
}
extern "C" int sfg_geno_colsums(sfg_ctx *ctx, const sfg_geno *g, double *sum_host, double *sqsum_host) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (g->packed) {        // setup-time statistic: through a dense copy
        sfg_geno *u = nullptr; SFG_TRY(sfg_geno_unpack(ctx, g, &u));
        const int rc = sfg_geno_colsums(ctx, u, sum_host, sqsum_host); sfg_geno_free(ctx, u); return rc;
    }
    SFG_TRY(sfg_ws_reserve(ctx, g->ncol * 16));
    double *ds = (double *)ctx->ws, *dq = ds + g->ncol;
    SFG_HIP(ctx, hipMemsetAsync(ds, 0, g->ncol * 16, ctx->stream));            // the row chunks are combined with (exact, integer-valued) fp64 atomics
    hipLaunchKernelGGL(k_colsums, dim3((unsigned)((g->ncol + 255) / 256), (unsigned)((g->nrow + CS_ROWS - 1) / CS_ROWS)), dim3(256), 0, ctx->stream, g->dev, g->nrow, g->ncol, g->ld, ds, dq);
    SFG_HIP(ctx, hipGetLastError());
    if (sum_host) SFG_HIP(ctx, hipMemcpyAsync(sum_host, ds, g->ncol * 8, hipMemcpyDeviceToHost, ctx->stream));
    if (sqsum_host) SFG_HIP(ctx, hipMemcpyAsync(sqsum_host, dq, g->ncol * 8, hipMemcpyDeviceToHost, ctx->stream));
    SFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
extern "C" int sfg_reduce_rows_dev(sfg_ctx *ctx, uint64_t *rows, size_t nrows_of_L, int L) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    const size_t rows_total = nrows_of_L * L;
    if (!rows_total) return 0;
    hipLaunchKernelGGL(k_reduce_rows, dim3((unsigned)(rows_total * (SFG_N / 256))), dim3(256), 0, ctx->stream, (u64 *)rows, L, ctx->modc);
    SFG_HIP(ctx, hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------- rotation cache of one operand block row
// rotc[baby][i] = RotateRight(A[i][bi], -baby) (matmult.go:1373-1377) for the active baby steps, then (LDS-DMA MAC) the fp64
// operand form into rotf_dst.  a_row / rotc are scratch of s resp. d*s ciphertexts.
static int build_rot_row_tab(sfg_ctx *ctx, const u64 *A, int s, int nl_in, int nl, int lev, int L, int nbr, int bi, const std::vector<uint8_t> &baby_t,
                             u64 *a_row, u64 *rotc, bool dma, double *rotf_dst);
static int build_rot_row(sfg_ctx *ctx, const u64 *A, int s, int nl_in, int nl, int lev, int L, const Shape &sh, int bi, u64 *a_row, u64 *rotc,
                         bool dma, double *rotf_dst) {
    const int d = SFG_D;
    const int nr = sh.rows_of(bi);
    // active baby steps (matmult.go:1326-1336), union over ALL block columns of the operand as in the reference
    std::vector<uint8_t> baby_t(d, 0);
    for (int shift = 0; shift < SFG_SLOTS; shift++) {
        if (baby_t[shift % d]) continue;
        bool any = false;
        for (int bj = 0; bj < sh.m_ct && !any; bj++) any = diag_bool(nr, sh.cols_of(bj), SFG_SLOTS, -shift);
        if (any) baby_t[shift % d] = 1;
    }
    return build_rot_row_tab(ctx, A, s, nl_in, nl, lev, L, sh.nbr, bi, baby_t, a_row, rotc, dma, rotf_dst);
}
static int build_rot_row_tab(sfg_ctx *ctx, const u64 *A, int s, int nl_in, int nl, int lev, int L, int nbr, int bi, const std::vector<uint8_t> &baby_t,
                             u64 *a_row, u64 *rotc, bool dma, double *rotf_dst) {
    const int N = SFG_N, d = SFG_D; const size_t ctw = (size_t)2 * nl * N;
    for (int i = 0; i < s; i++) {                  // A[i][bi] at the dropped level, contiguous over i
        const u64 *src = A + ((size_t)i * nbr + bi) * 2 * nl_in * N;
        if (nl == nl_in) SFG_HIP(ctx, hipMemcpyAsync(a_row + (size_t)i * ctw, src, ctw * 8, hipMemcpyDeviceToDevice, ctx->stream));
        else hipLaunchKernelGGL(k_drop_level, dim3((unsigned)(2 * nl * (N / 256))), dim3(256), 0, ctx->stream, src, a_row + (size_t)i * ctw, nl_in, nl);
    }
    SFG_HIP(ctx, hipGetLastError());
    {
        PhaseTimer t(ctx, "rotate");
        std::vector<int> nrv((size_t)d * s, 0), inv((size_t)d * s, 0);
        for (int baby = 0; baby < d; baby++) for (int i = 0; i < s; i++) { nrv[(size_t)baby * s + i] = baby_t[baby] ? -baby : 0; inv[(size_t)baby * s + i] = i; }
        // LDS-DMA / broadcast MAC: the rotated ciphertexts are produced directly as its fp64 operand rows (no u64 rotation cache, no conversion pass)
        if (dma) SFG_TRY(launch_rotate_right_indexed_f64(ctx, a_row, s, rotf_dst, d * s, lev, nrv.data(), inv.data(), L));
        else SFG_TRY(launch_rotate_right_indexed(ctx, a_row, s, rotc, d * s, lev, nrv.data(), inv.data()));
        t.stop(1);
    }
    return 0;
}

// ---------------------------------------------------------------- phase 1: accumulate
// acc_dev: [(j - j0)][giant < d][i < s][2][L][N] canonical residues (zero-initialised here unless accumulate != 0)
// for operand block rows [b0, b1) and block columns [j0, j1).
static int matmul_accumulate(sfg_ctx *ctx, const u64 *A, int s, int in_level, int max_level, const Shape &sh, unsigned flags,
                             int b0, int b1, int j0, int j1, int accumulate, u64 *acc, const double *rotf_pre = nullptr, const double *rotsum_pre = nullptr,
                             const I8RotPre *pre8 = nullptr, size_t acc_col_words = 0) {          // acc_col_words: words between the accumulators of consecutive block columns (0: dense, 91 giants)
    const int N = SFG_N, d = SFG_D, L = max_level;
    if (pre8 && (rotf_pre || !mac_use_dma(ctx) || b0 != 0 || b1 != pre8->nbr || sh.nbr != pre8->nbr || s != pre8->s || L != pre8->L))
        SFG_FAIL(ctx, "matmul: internal: the int8 rot tiles were built for another product (block rows %d, s = %d, level %d)", pre8->nbr, pre8->s, pre8->L);
    const bool rot_ext = rotf_pre || pre8;                  // the caller holds the rotations of every block row
    const int lev = in_level > max_level ? max_level : in_level, nl = lev + 1, nl_in = in_level + 1;
    if (L < 1 || L > ctx->nq) SFG_FAIL(ctx, "matmul: max_level out of range");
    if (nl < L) SFG_FAIL(ctx, "matmul: input level %d has fewer than max_level = %d moduli", in_level, max_level);
    if (b0 < 0 || b1 > sh.nbr || b0 > b1 || j0 < 0 || j1 > sh.m_ct || j0 > j1) SFG_FAIL(ctx, "matmul: block range out of bounds");
    const size_t ctw = (size_t)2 * nl * N, accw = (size_t)s * 2 * L * N;
    const int ncolb = j1 - j0;
    const size_t acc_col = acc_col_words ? acc_col_words : (size_t)d * accw;
    if (acc_col < (size_t)d * accw) SFG_FAIL(ctx, "matmul: internal: accumulator column stride below 91 giant steps");
    if (b0 == b1 || j0 == j1) {
        if (!accumulate) for (int c = 0; c < ncolb; c++) SFG_HIP(ctx, hipMemsetAsync(acc + (size_t)c * acc_col, 0, (size_t)d * accw * 8, ctx->stream));
        return 0;
    }
    const bool dma = mac_use_dma(ctx);                          // LDS-DMA MAC: half-row plaintexts, fp64 rot operand, block-row groups
    const size_t prow = dma ? (size_t)N / 2 : (size_t)N;     // words per plaintext modulus row
    // G block rows share one MAC launch (K = G*91): accumulators are written once per group instead of
    // read-modify-written per block.  Bounded by scratch: ~4.9 GB per block row at s = 15.
    int G = 1;
    const size_t nplain = (size_t)d * d;                     // 8281 >= 8192 slots per block row: the tail stays zero
    if (dma) {
        G = ctx->cfg.mm_group;
        // 16 (24) block rows per launch halve (third) the accumulator read-modify-writes and the per-launch prologues (16: -1.7 % at 100k x 1M, identical bits) but
        // need a 43 (65) GB plaintext panel and, for the pipelined rotation caches, 2 x 34.5 (52) GB of operands: taken only when that fits beside what is resident
        // (not for a single block column against a caller's rotation cache - the association scan: fewer launches save a few accumulator passes there, and the
        //  larger panel competes with the 115 GB cache for HBM: measured 0.49 s instead of 0.31 s per batch)
        if (pre8) G = pre8->G;
        else if (ctx->cfg.mm_group_auto && b1 - b0 > G && (j1 - j0 >= 4 || !rotf_pre)) {
            std::vector<int> po, ib; const int npl = mac_dma_planes(ctx, L, po, ib);
            size_t have = 0, total = 0;
            if (npl > 0 && hipMemGetInfo(&have, &total) == hipSuccess) {
                for (const auto &kv : ctx->pool) if (kv.first == "mm.pt" || kv.first == "mm.rotf" || kv.first.rfind("mi8.", 0) == 0) have += kv.second.second;   // regrown in place
                for (int cand : {24, 20, 16, 14, 12, 10}) {
                    const int G2 = std::min(cand, b1 - b0);
                    if (G2 <= G) break;
                    const bool pipe2 = !rotf_pre && b1 - b0 > G2 && !ctx->cfg.no_overlap;
                    const bool enc2 = !ctx->cfg.no_overlap && !ctx->cfg.no_enc_overlap && (size_t)((b1 - b0 + G2 - 1) / G2) * (j1 - j0) >= 2;
                    const bool streamable2 = ctx->cfg.mac_i8 && ctx->cfg.mac_i8_big && ctx->cfg.mac_i8_ring && ctx->cfg.stage_pack && !enc2;      // then the panel holds 4 block rows
                    const bool ride2 = ctx->cfg.mac_i8 && ctx->cfg.pt_ride > 0 && !enc2 && !streamable2 && (j1 - j0 >= 2 || (pre8 && b1 - b0 > G2));      // (the riding transposition keeps two panels too)
                    size_t ptb = (size_t)L * ((size_t)N / 2) * 8;                    // bytes per plaintext: compact rows where every modulus is on the int8 MAC (as decided below)
                    if (ctx->cfg.pt_compact && ctx->cfg.mac_i8 && !streamable2 && mac_dma_packed_mask(ctx, L)) {
                        bool big_ok = true; size_t planes = 0;
                        for (int l = 0; l < L; l++) { const bool sm = ctx->q[l] < (1ULL << 36); planes += sm ? 5 : 6; if (!sm && !ctx->cfg.mac_i8_big) big_ok = false; }
                        if (big_ok && ((b1 - b0 + G2 - 1) / G2 <= 2 || j1 - j0 >= 4)) ptb = planes * ((size_t)N / 2);
                    }
                    size_t need = (size_t)(streamable2 ? std::min(G2, 4) : G2) * nplain * ptb * (enc2 || ride2 ? 2 : 1);
                    if (!rotf_pre) need += ((size_t)G2 * d + 3) * s * 2 * (size_t)npl * N * 8 * (pipe2 ? 2 : 1);
                    if (ctx->cfg.mac_i8) {                         // + the two operand streams and the tile-ordered results of the int8 MAC (small moduli)
                        int nsm = 0; for (int l = 0; l < L; l++) nsm += ctx->q[l] < (1ULL << 36);
                        need += mac_i8_stream_bytes(G2 * d, nsm, 5, 1);          // (one transposed rot copy: a group's copy is recycled for the next group, in stream order)
                        if (ctx->cfg.mac_i8_big && nsm < L) need += mac_i8_stream_bytes(G2 * d, 1, 6, 1);
                    }
                    if (need + (12ULL << 30) <= have) { G = G2; break; }
                }
            }
        }
        if (G > b1 - b0) G = b1 - b0;
    }
    u64 *a_row = nullptr, *rotc = nullptr, *pt = nullptr; int8_t *skew = nullptr; double *rotf = nullptr, *rotsum = nullptr; size_t rowf = 0;
    const unsigned packed_mask = dma ? mac_dma_packed_mask(ctx, L) : 0u;      // small-modulus plaintext rows in the packed-limb format
    // The int8 MAC multiplies with a k-contiguous copy of a group's rot operand, transposed when the operand changes and kept for two operands.  That pays when
    // the copy is reused: several block columns in this call, or so few groups that the copies survive from call to call (a caller's rotation cache multiplied one
    // block column at a time).  The association scan - one block column per batch against a 62-block-row cache - takes the fp64 kernel.
    bool keep_all = false;                                  // a caller's rotation cache of up to 16 groups whose transposed copies all fit: the association scan
    if (dma && ctx->cfg.mac_i8 && packed_mask && rotf_pre && (b1 - b0 + G - 1) / G <= 16) {
        int nsm = 0; for (int l = 0; l < L; l++) nsm += ctx->q[l] < (1ULL << 36);
        const size_t per = mac_i8_stream_bytes(G * d, nsm, 5, 1) - mac_i8_stream_bytes(G * d, nsm, 5, 0), ngr = (size_t)(b1 - b0 + G - 1) / G;
        size_t fr = 0, tot = 0, held = 0;
        for (const auto &kv : ctx->pool) if (kv.first.rfind("mi8.A", 0) == 0) held += kv.second.second;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) keep_all = fr + held >= ngr * per + mac_i8_stream_bytes(G * d, nsm, 5, 0) + ctx->cfg.i8_keep_reserve;     // (the panel, accumulators and key-switch scratch of the call are still to be allocated the first time)
    }
    const bool use_i8 = pre8 || (dma && ctx->cfg.mac_i8 && packed_mask && ((b1 - b0 + G - 1) / G <= 2 || j1 - j0 >= 4 || keep_all));
    const bool use_i8_big = pre8 || (use_i8 && ctx->cfg.mac_i8_big);                    // the 46-bit modulus too: six digit planes, its own pair of transposed rot copies
    const size_t grp_slices = (size_t)G * d + 3;            // k-slices of one group's fp64 rotation cache (+ 3: see launch_mac_dma)
    const bool pipelined = dma && !rot_ext && b1 - b0 > G && !ctx->cfg.no_overlap;
    SFG_TRY(sfg_scratch(ctx, "mm.a_row", (size_t)s * ctw * 8, (void **)&a_row));
    SFG_TRY(sfg_scratch(ctx, "mm.rotc", dma ? 8 : (size_t)d * s * ctw * 8, (void **)&rotc));     // u64 rotation cache: only the register-staged MAC reads one
    // Two plaintext panels when the encode of launch k + 1 runs on its own queue beside the transposition + MAC of launch k (fp64-issue bound beside HBM bound)
    const bool enc_ov = dma && !ctx->cfg.no_overlap && !ctx->cfg.no_enc_overlap && (size_t)((b1 - b0 + G - 1) / G) * (j1 - j0) >= 2;
    // where every modulus multiplies on the int8 matrix core from streamed tiles, the panel serves only the rare launches that cannot stream: Gp block rows of it
    const bool streamable = use_i8 && use_i8_big && ctx->cfg.mac_i8_ring && ctx->cfg.stage_pack && !enc_ov && !pre8;
    const int Gp = streamable ? std::min(G, 4) : G;
    // Compact panel rows (round 6): where every modulus of the product multiplies on the int8 matrix core the panel holds nothing but digit planes - five (six) planes of
    // N/2 bytes per modulus, back to back: 208 KiB per plaintext at L = 5 instead of five rows of N/2 words (320 KiB).  Room for the second panel of the riding transposition.
    bool all_small = true; for (int l = 0; l < L; l++) if (ctx->q[l] >= (1ULL << 36)) all_small = false;
    const bool compact = ctx->cfg.pt_compact && dma && use_i8 && (use_i8_big || all_small) && !streamable;
    size_t plw = (size_t)L * prow;                           // words per plaintext
    int pt_planes = 0;
    if (compact) { for (int l = 0; l < L; l++) pt_planes += ctx->q[l] < (1ULL << 36) ? 5 : 6; plw = (size_t)pt_planes * ((size_t)N / 2) / 8; }
    // K-major panel (round 6): the compact panel's bytes ordered [column][plane][128-byte coefficient block][k][128 B], so that the 16 k of a transposition unit's
    // column are one 2 KiB run (the NTT's stores are 128-byte runs either way).  Whole-group launches only (a sub-launch would change K between encode and MAC).
    const bool kmajor = compact && ctx->cfg.pt_kmajor && Gp == G && (size_t)G * d * 128 * 64 * 32 < (1ULL << 31);
    const int pt_layout = kmajor ? 2 : compact ? 1 : 0;
    const size_t panel_words = (size_t)Gp * nplain * plw;
    // The riding transposition (kernels.hpp PtRide): the panel of MAC launch k - 1 is transposed by mover workgroups inside the plaintext-NTT launches of launch k's
    // encode, which writes the OTHER panel; MAC launch k - 1 follows that encode on the same queue and finds its tiles in place.  Taken where every modulus multiplies
    // on the int8 matrix core from digit-plane panels; the first launch after a change of block-row group (its rot operand's buffer is about to be rebuilt) and the
    // call's last launch transpose by the pass as before.
    // (a launch rides in the encode of the NEXT block column of its group - or of the next group's first column where the rot tiles of every group are the caller's,
    //  I8RotPre: nothing is rebuilt between groups then, so the multi-GPU engine's one-column calls over several groups ride as well)
    bool ride_want = use_i8 && !streamable && !enc_ov && ctx->cfg.pt_ride > 0 && (j1 - j0 >= 2 || (pre8 && b1 - b0 > G)) &&
                     (use_i8_big || [&] { for (int l = 0; l < L; l++) if (ctx->q[l] >= (1ULL << 36)) return false; return true; }());
    // (+ 64 KiB: the transposition walks whole chunks of 64 k, and in the K-major panel the rows K .. K + 63 of the last column's last coefficient block - read, then
    //  masked - lie up to 8 KiB past the panel)
    // A caller's group size (SFG_MM_GROUP, sfg_config.mm_group) may leave room for one panel only: the product then transposes by the pass, as before round 6
    if (ride_want && sfg_scratch(ctx, "mm.pt", panel_words * 8 * 2 + 65536, (void **)&pt)) { ctx->err.clear(); ride_want = false; }
    if (!ride_want) SFG_TRY(sfg_scratch(ctx, "mm.pt", panel_words * 8 * (enc_ov ? 2 : 1) + 65536, (void **)&pt));
    u64 *const pt_base = pt;
    SFG_TRY(sfg_scratch(ctx, "mm.skew", (size_t)SFG_SLOTS * SFG_SLOTS, (void **)&skew));
    int8_t *unpacked = nullptr;
    if (sh.g->packed) SFG_TRY(sfg_scratch(ctx, "mm.unpack", (size_t)SFG_SLOTS * SFG_SLOTS, (void **)&unpacked));
    if (dma) {
        std::vector<int> plane_of, is_big; const int nplanes = mac_dma_planes(ctx, L, plane_of, is_big);
        if (nplanes < 0) return 1;
        rowf = (size_t)nplanes * N;
        if (!rot_ext) SFG_TRY(sfg_scratch(ctx, "mm.rotf", grp_slices * s * 2 * rowf * 8 * (pipelined ? 2 : 1), (void **)&rotf));
        if (!rot_ext && packed_mask) SFG_TRY(sfg_scratch(ctx, "mm.rotsum", (size_t)2 * s * 2 * rowf * 8, (void **)&rotsum));     // one per ring half
    }
    int rc = 0;
    bool first_group = true;
    // rotation cache of one group into half `buf` of mm.rotf (on whatever stream is current)
    auto build_group = [&](int bg, int buf) -> int {
        const int ng = std::min(G, b1 - bg);
        ctx->i8_gen++;                                          // (the int8 MAC keeps a transposed copy per rot operand: this one changes now)
        double *dst = rotf + (size_t)buf * grp_slices * s * 2 * rowf;
        for (int g = 0; g < ng; g++) SFG_TRY(build_rot_row(ctx, A, s, nl_in, nl, lev, L, sh, bg + g, a_row, rotc, dma, dma ? dst + (size_t)g * d * s * 2 * rowf : nullptr));
        if (dma && (ng * d) % 4)          // the ragged last MAC chunk reads up to 3 k-slices past the group against zero plaintexts: keep them finite
            SFG_HIP(ctx, hipMemsetAsync(dst + (size_t)ng * d * s * 2 * rowf, 0, (size_t)3 * s * 2 * rowf * 8, ctx->stream));
        if (packed_mask) SFG_TRY(launch_rot_sum(ctx, dst, (size_t)s * 2, ng * d, L, rotsum + (size_t)buf * s * 2 * rowf));
        return 0;
    };
    // With several groups the key switching of group k+1 runs on the auxiliary stream beside the encode + MAC of group k
    // (those kernels leave registers and wave slots free; the MAC does not, so nothing overlaps it).
    hipStream_t main_stream = ctx->stream;
    if (pipelined) {
        SFG_TRY(sfg_stream_after(ctx, ctx->aux_stream, main_stream));            // inputs and scratch as the main stream left them
        { AuxScope aux(ctx); SFG_TRY(build_group(b0, 0)); }
        SFG_HIP(ctx, hipEventRecord(ctx->ev_pipe[0], ctx->aux_stream));
    }
    int gi = 0, it = 0;
    // a MAC launch: the plaintext panel `ptp` of `gsn` block rows against the group's rot operand, into block column accumulator `accj`
    struct MacJob { bool on = false; u64 *ptp = nullptr; int gsn = 0, sub0 = 0, gi = 0, acc_flag = 0; u64 *accj = nullptr; const double *rotf_grp = nullptr, *rotsum_grp = nullptr; StagePack *sp = nullptr; };
    int8_t *rideBs = nullptr, *rideBb = nullptr;               // the riding launches' tile buffers (a launch of the same call that transposes by the pass uses them too)
    auto run_mac = [&](const MacJob &m, int B_mode) -> int {
        PhaseTimer t(ctx, "mac");
        MacStrides st;
        if (m.sp) { st.B_small = m.sp->Bs; st.B_big = m.sp->Bb; st.kb = m.sp->kb; }
        else if (rideBs) { st.B_small = rideBs; st.B_big = rideBb; st.B_mode = B_mode; }
        if (pre8) { st.A_small = pre8->As[m.gi]; st.A_big = pre8->Ab[m.gi]; }
        st.rot_k = (size_t)s * ctw; st.rot_r = (size_t)nl * N;          // rotc[baby][i][poly][nl][N]: row r = i*2+poly
        st.pt_k = plw; st.pt_n = (size_t)m.gsn * d * plw; st.pt_half = dma; st.pt_packed = packed_mask != 0; st.pt_digits = st.i8 = use_i8; st.i8_big = st.pt_digits_big = use_i8_big;   // pt[giant][g][baby]: k = g*91 + baby
        st.pt_layout = pt_layout; st.pt_L = L;
        st.out_n = accw; st.out_r = (size_t)L * N;                      // acc[j][giant][r]
        int r2;
        if (dma) r2 = launch_mac_dma(ctx, pre8 ? nullptr : m.rotf_grp + (size_t)m.sub0 * d * s * 2 * rowf, (size_t)s * 2, m.ptp, m.accj, m.gsn * d, 2 * s, d, L, m.acc_flag, st, m.rotsum_grp);
        else r2 = launch_mac_strided(ctx, rotc, m.ptp, m.accj, d, 2 * s, d, L, m.acc_flag, st);
        t.stop(1);
        return r2;
    };
    MacJob held;                                               // the delayed MAC launch whose panel the next encode's NTT launches transpose
    struct StreamRestore { sfg_ctx *c; hipStream_t s; ~StreamRestore() { c->stream = s; } } restore_main{ctx, main_stream};       // whatever path leaves the loop
    if (enc_ov) SFG_TRY(sfg_stream_after(ctx, ctx->enc_stream, main_stream));      // the genotypes, the cache slots and whatever the caller enqueued before
    for (int bg = b0; bg < b1 && !rc; bg += G, gi++) {
        const int ng = std::min(G, b1 - bg);
        // (the delayed launch of the previous group reads a rot operand buffer that is rebuilt below: it goes first, transposing by the pass - unless the rot tiles
        //  of all groups are the caller's)
        if (held.on && !pre8) { rc = run_mac(held, 2); held.on = false; if (rc) break; }
        // ---- rotation caches of the group's block rows (or the product-wide cache built by the caller)
        const double *rotf_grp = rotf, *rotsum_grp = rotsum;
        if (rotf_pre) { rotf_grp = rotf_pre + (size_t)(bg - b0) * d * s * 2 * rowf; rotsum_grp = rotsum_pre ? rotsum_pre + (size_t)gi * s * 2 * rowf : nullptr; }
        else if (pre8) { rotf_grp = nullptr; rotsum_grp = nullptr; }            // group gi multiplies from pre8->As[gi] / Ab[gi]
        else if (pipelined) {
            if (bg + G < b1) {                                                   // next group: its half was last read by group gi-1
                SFG_TRY(sfg_stream_after(ctx, ctx->aux_stream, main_stream));
                { AuxScope aux(ctx); SFG_TRY(build_group(bg + G, (gi + 1) & 1)); }
                SFG_HIP(ctx, hipEventRecord(ctx->ev_pipe[(gi + 1) & 1], ctx->aux_stream));
            }
            SFG_HIP(ctx, hipStreamWaitEvent(main_stream, ctx->ev_pipe[gi & 1], 0));
            rotf_grp = rotf + (size_t)(gi & 1) * grp_slices * s * 2 * rowf;
            if (rotsum) rotsum_grp = rotsum + (size_t)(gi & 1) * s * 2 * rowf;
        } else {
            rc = build_group(bg, 0);
            if (rc) break;
        }
        for (int bj = j0; bj < j1 && !rc; bj++, it++) {
            const int nc = sh.cols_of(bj);
            // Streamed transposition (StagePack, kernels.hpp): when every block of this launch has all 8192 diagonals and every modulus multiplies on the int8
            // matrix core, the NTT's digit planes go batch by batch through a cache-resident staging buffer into the MAC's tiles - no panel, no transposition pass
            bool stream = streamable;
            for (int g = 0; g < ng && stream; g++) stream = sh.rows_of(bg + g) + nc > SFG_SLOTS;
            StagePack sp;
            if (stream) {
                std::vector<int> po, ib; (void)mac_dma_planes(ctx, L, po, ib);
                int nbig = 0; sp.l_big = -1; sp.l_small0 = -1; sp.n_small = 0;
                for (int l = 0; l < L; l++) { if (ib[l]) { nbig++; sp.l_big = l; } else { if (sp.l_small0 < 0) sp.l_small0 = l; sp.n_small++; } }
                bool contiguous = nbig <= 1;
                for (int l = 0; l < L && contiguous; l++) if (!ib[l] && (l < sp.l_small0 || l >= sp.l_small0 + sp.n_small)) contiguous = false;
                if (!contiguous || !sp.n_small) stream = false;
            }
            if (stream) {
                sp.kb = 92; sp.njt = 6; sp.nch = (ng * sp.kb + 63) / 64; sp.q = ctx->cfg.stage_same_queue ? ctx->stream : ctx->enc_stream; sp.ev_ntt = ctx->ev_enc[0]; sp.ev_pack = ctx->ev_enc[1];
                const size_t nBs = mac_i8_tile_bytes(ng * sp.kb, sp.n_small, 5), nBb = sp.l_big >= 0 ? mac_i8_tile_bytes(ng * sp.kb, 1, 6) : 0;
                const size_t had_s = ctx->pool.count("mi8.Bs") ? ctx->pool["mi8.Bs"].second : 0, had_b = ctx->pool.count("mi8.Bb") ? ctx->pool["mi8.Bb"].second : 0;
                rc = sfg_scratch(ctx, "mi8.Bs", nBs, (void **)&sp.Bs); if (rc) break;
                if (nBb) { rc = sfg_scratch(ctx, "mi8.Bb", nBb, (void **)&sp.Bb); if (rc) break; }
                rc = sfg_scratch(ctx, "mi8.stage", (size_t)ctx->cfg.stage_giants * SFG_D * L * (N / 2) * 8, (void **)&sp.stage); if (rc) break;
                // what no batch owns (columns 91..95, k' past the group's last block row) must read as zero: cleared when the buffers are new or the group shape changes
                if (had_s < nBs || had_b < nBb || ctx->sp_shape != ng) {
                    SFG_HIP(ctx, hipMemsetAsync(sp.Bs, 0, nBs, ctx->stream));
                    if (nBb) SFG_HIP(ctx, hipMemsetAsync(sp.Bb, 0, nBb, ctx->stream));
                    ctx->sp_shape = ng;
                }
                // the transposition queue starts behind everything this queue has done (the previous MAC launch read the tiles, the memsets above)
                rc = sfg_stream_after(ctx, sp.q, ctx->stream); if (rc) break;
            }
            const int pbuf = enc_ov || ride_want ? (it & 1) : 0;
            pt = pt_base + (size_t)pbuf * panel_words;
            if (enc_ov) {                                  // encode on its queue: after the MAC that last read this panel buffer
                if (it >= 2) SFG_HIP(ctx, hipStreamWaitEvent(ctx->enc_stream, ctx->ev_enc[2 + pbuf], 0));
                ctx->stream = ctx->enc_stream;
            }
          // A launch that cannot stream (a block with fewer than 8192 diagonals: the corner of a ragged matrix) goes through the plaintext panel.  Where streaming
          // is the rule the panel holds only Gp block rows, and such a launch is multiplied in sub-launches of Gp rows that accumulate onto each other.
          const int step = stream ? ng : std::min(ng, Gp);
          // riding: this launch's encode carries the held launch's transposition; its own MAC is held in turn (whole-group launches only)
          const bool ride_this = ride_want && !stream && step == ng;
          PtRide ride;
          if (ride_this && !rideBs) { rc = i8_ride_tiles(ctx, G * d, L, &rideBs, &rideBb); if (rc) break; }
          if (held.on && ride_this && rideBs) {
              int launches = 0;
              for (int g = 0; g < ng; g++) {
                  const int nr = sh.rows_of(bg + g);
                  if (nr + nc > SFG_SLOTS) launches += encode_rows_launches(ctx, SFG_SLOTS);
                  else launches += encode_rows_launches(ctx, nr) + (nc > 1 ? encode_rows_launches(ctx, nc - 1) : 0);
              }
              rc = i8_ride_prepare(ctx, held.ptp, held.gsn * d, d, plw, (size_t)held.gsn * d * plw, pt_layout, L, launches, ride); if (rc) break;
              if (ride.on && (ride.job.a5.B != rideBs || (ride.job.n6 && ride.job.a6.B != rideBb))) { rc = 1; ctx->err = "matmul: internal: the riding transposition's tile buffers moved"; break; }
          }
          if (held.on && !ride.on) { rc = run_mac(held, rideBs ? 2 : 0); held.on = false; if (rc) break; }       // nothing to ride in: the held launch goes now, by the pass
          for (int sub0 = 0; sub0 < ng && !rc; sub0 += step) {
            const int gs = std::min(step, ng - sub0);
            for (int g = sub0; g < sub0 + gs && !rc; g++) {
                const int bi = bg + g, nr = sh.rows_of(bi);
                // plaintext coefficient cache of the stored block (sfg_geno_set_plaintext_cache)
                PcCache pcc; uint64_t ptc_key = 0; bool ptc_new = false;
                if (dma && sh.g->ptc_budget && sh.g->ptc_owner == (const void *)ctx) {
                    const uint64_t sr = sh.transposed ? bj : bi, sc = sh.transposed ? bi : bj;
                    ptc_key = sr << 33 | sc << 1 | ((flags & SFG_SQUARE) ? 1u : 0u);
                    const size_t slot_bytes = (size_t)SFG_SLOTS * SFG_SLOTS * 8;
                    auto it = sh.g->ptc.find(ptc_key);
                    if (it != sh.g->ptc.end()) { pcc.slot = it->second.slot; pcc.mode = it->second.transposed == sh.transposed ? 2 : 3; pcc.perm = sh.g->ptc_perm; sh.g->ptc_hits++; }
                    else if (sh.g->ptc_used + slot_bytes <= sh.g->ptc_budget) {
                        double *slot = sh.g->ptc_arena + sh.g->ptc_used / 8;
                        sh.g->ptc[ptc_key] = sfg_geno::PtcEntry{slot, sh.transposed}; sh.g->ptc_used += slot_bytes; sh.g->ptc_fills++;
                        pcc.slot = slot; pcc.mode = 1; ptc_new = true;
                    }
                }
                // (a failed fill is the newest slot of the arena: give it back)
                auto ptc_undo = [&]() { if (ptc_new) { auto it = sh.g->ptc.find(ptc_key); if (it != sh.g->ptc.end()) { sh.g->ptc.erase(it); sh.g->ptc_used -= (size_t)SFG_SLOTS * SFG_SLOTS * 8; } ptc_new = false; } };
                if (pcc.mode < 2) {
                    PhaseTimer t(ctx, "skew");
                    if (sh.g->packed) {        // expand the stored block (rows x cols as stored) into the int8 staging block, then skew as usual
                        const size_t sr0 = (size_t)(sh.transposed ? bj : bi) * SFG_SLOTS, sc0 = (size_t)(sh.transposed ? bi : bj) * SFG_SLOTS;
                        rc = launch_geno_unpack(ctx, sh.g, sr0, sc0, sh.transposed ? nc : nr, sh.transposed ? nr : nc, unpacked, SFG_SLOTS);
                        if (!rc) rc = launch_skew(ctx, unpacked, SFG_SLOTS, nr, nc, sh.transposed ? 1 : 0, (flags & SFG_SQUARE) ? 1 : 0, skew);
                    } else rc = launch_skew(ctx, sh.block(bi, bj), sh.ld, nr, nc, sh.transposed ? 1 : 0, (flags & SFG_SQUARE) ? 1 : 0, skew);
                    t.stop(1);
                }
                if (rc) { ptc_undo(); break; }
                // existing diagonals of this block form at most two runs of shifts: [0, nr) and (n - nc, n)  (GetDiagBool)
                int runs[2][2]; int nruns = 0;
                if (nr + nc > SFG_SLOTS) { runs[0][0] = 0; runs[0][1] = SFG_SLOTS; nruns = 1; }
                else { runs[0][0] = 0; runs[0][1] = nr; runs[1][0] = SFG_SLOTS - nc + 1; runs[1][1] = SFG_SLOTS; nruns = runs[1][0] < runs[1][1] ? 2 : 1; }
                const bool full = nruns == 1 && runs[0][1] - runs[0][0] == SFG_SLOTS;
                // zero what the encoder will not write: plaintext slot of (giant, g, baby) is ((giant*ng + g)*91 + baby)
                if (stream) {      // (the tiles' unowned positions are zero already, owned ones without a plaintext are written as zeros)
                } else if (kmajor) {      // (K-major panel: rows of block row g in column 90 past shift 8191, or in every column)
                    const int kb0 = (g - sub0) * d;
                    if (full) rc = launch_pt_zero_kmajor(ctx, pt, gs * d, pt_planes, d - 1, d, kb0 + (SFG_SLOTS - (d - 1) * d), kb0 + d);
                    else rc = launch_pt_zero_kmajor(ctx, pt, gs * d, pt_planes, 0, d, kb0, kb0 + d);
                } else if (full) {        // only the 89 slots past shift 8191 (giant 90, baby 2..90)
                    rc = launch_pt_zero(ctx, pt + (((size_t)(d - 1) * gs + (g - sub0)) * d + (SFG_SLOTS - (d - 1) * d)) * plw, 0, (nplain - SFG_SLOTS) * plw, 1, L, prow, packed_mask);
                } else {           // ragged block: all 91 x 91 slots of this block row
                    rc = launch_pt_zero(ctx, pt + (size_t)(g - sub0) * d * plw, (size_t)gs * d * plw, (size_t)d * plw, d, L, prow, packed_mask);
                }
                if (rc) { ptc_undo(); break; }
                {
                    PhaseTimer t(ctx, "encode");
                    for (int r = 0; r < nruns && !rc; r++) {
                        if (stream) sp.g = g;
                        if (dma) rc = launch_encode_rows(ctx, skew, runs[r][0], runs[r][1] - runs[r][0], L, pt, true, gs, g - sub0, packed_mask | (use_i8 ? 0x80000000u : 0u) | (use_i8_big ? 0x40000000u : 0u) | (compact ? PT_COMPACT : 0u) | (kmajor ? PT_KMAJOR : 0u),
                                                         pcc.mode ? &pcc : nullptr, stream ? &sp : nullptr, ride.on ? &ride : nullptr);
                        else rc = launch_encode_rows(ctx, skew, runs[r][0], runs[r][1] - runs[r][0], L, pt + (size_t)runs[r][0] * plw, false);
                    }
                    t.stop(nruns);
                }
                if (rc) ptc_undo();
            }
            if (enc_ov) {
                ctx->stream = main_stream;
                if (!rc) { SFG_HIP(ctx, hipEventRecord(ctx->ev_enc[pbuf], ctx->enc_stream)); SFG_HIP(ctx, hipStreamWaitEvent(main_stream, ctx->ev_enc[pbuf], 0)); }
            }
            if (rc) break;
            if (!stream && streamable) ctx->sp_shape = -1;       // (this launch transposes through the tile buffers the streamed launches keep partly cleared)
            if (stream && sp.pending) SFG_HIP(ctx, hipStreamWaitEvent(ctx->stream, sp.ev_pack, 0));       // the last batch is in the tiles
            // the held launch: whatever of its transposition no NTT launch of this encode took, then its MAC on the tiles
            if (held.on) { rc = i8_ride_finish(ctx, ride); if (!rc) rc = run_mac(held, 1); held.on = false; if (rc) break; }
            {
                MacJob m; m.on = true; m.ptp = pt; m.gsn = gs; m.sub0 = sub0; m.gi = gi; m.sp = stream ? &sp : nullptr;
                m.acc_flag = (accumulate || !first_group || sub0 > 0) ? 1 : 0;      // the first (sub-)launch of a fresh call overwrites
                m.accj = acc + (size_t)(bj - j0) * acc_col; m.rotf_grp = rotf_grp; m.rotsum_grp = rotsum_grp;
                if (ride_this && rideBs) held = m;                                 // multiplied after the next launch's encode has transposed this panel
                else rc = run_mac(m, rideBs ? 2 : 0);
            }
          }
            if (enc_ov && !rc) SFG_HIP(ctx, hipEventRecord(ctx->ev_enc[2 + pbuf], main_stream));
        }
        first_group = false;
    }
    if (held.on && !rc) rc = run_mac(held, 2);                  // the call's last launch: nothing follows to ride in
    return rc;
}

// ---------------------------------------------------------------- phase 2: finalize
// out[i][j][2][L][N] (+)= sum_{giant in [g0,g1)} RotateRight(acc[j][giant][i], -giant*d)   (matmult.go:1443-1502)
// acc: [ncolb][acc_giants][s][2][L][N]; slot g of a block column holds giant step giant_base + g
static int matmul_finalize(sfg_ctx *ctx, const u64 *acc, int s, int max_level, int ncolb, int m_ct_out, int jout0, int g0, int g1,
                           const std::vector<uint8_t> *giant_active, int accumulate, u64 *out, int acc_giants = SFG_D, int giant_base = 0) {
    const int N = SFG_N, d = SFG_D, L = max_level;
    const size_t accw = (size_t)s * 2 * L * N, ctw = (size_t)2 * L * N;
    if (g0 < 0 || g1 > acc_giants || g0 > g1 || giant_base < 0) SFG_FAIL(ctx, "finalize: giant range out of bounds");
    u64 *rot = nullptr;
    SFG_TRY(sfg_scratch(ctx, "mm.fin_rot", (size_t)d * accw * 8, (void **)&rot));
    std::vector<int> glist;
    for (int g = g0; g < g1; g++) if (giant_base + g < d && (!giant_active || (*giant_active)[giant_base + g])) glist.push_back(g);
    const int ng = (int)glist.size();
    // only the listed giants are aligned (a rank of a giant-sharded finalize owns ~91/world of them): job (k, i) reads
    // accumulator ciphertext glist[k]*s + i and lands compactly at k*s + i
    std::vector<int> nrv((size_t)ng * s), inv((size_t)ng * s);
    for (int k = 0; k < ng; k++) for (int i = 0; i < s; i++) { nrv[(size_t)k * s + i] = -(giant_base + glist[k]) * d; inv[(size_t)k * s + i] = glist[k] * s + i; }
    int rc = 0;
    for (int jb = 0; jb < ncolb && !rc; jb++) {
        const u64 *accj = acc + (size_t)jb * acc_giants * accw;
        u64 *o = out + (size_t)(jout0 + jb) * ctw;
        if (ng) {
            PhaseTimer t(ctx, "rotate");
            rc = launch_rotate_right_indexed(ctx, accj, acc_giants * s, rot, ng * s, L - 1, nrv.data(), inv.data());
            t.stop(1);
            if (rc) break;
        }
        hipLaunchKernelGGL(k_sum_giants, dim3((unsigned)((size_t)s * 2 * L * (N / 256))), dim3(256), 0, ctx->stream, rot, ng, s, L,
                           o, (size_t)m_ct_out * ctw, accumulate, ctx->modc);
        if (hipGetLastError() != hipSuccess) { rc = 1; ctx->err = "finalize: k_sum_giants launch failed"; }
    }
    return rc;
}

static void giant_table(const Shape &sh, std::vector<uint8_t> &giant_t) {
    giant_t.assign(SFG_D, 0);
    for (int bi = 0; bi < sh.nbr; bi++) {
        const int nr = sh.rows_of(bi);
        for (int shift = 0; shift < SFG_SLOTS; shift++) {
            if (giant_t[shift / SFG_D]) { shift = (shift / SFG_D + 1) * SFG_D - 1; continue; }
            bool any = false;
            for (int bj = 0; bj < sh.m_ct && !any; bj++) any = diag_bool(nr, sh.cols_of(bj), SFG_SLOTS, -shift);
            if (any) giant_t[shift / SFG_D] = 1;
        }
    }
}

extern "C" int sfg_matmul_accumulate_dev(sfg_ctx *ctx, const uint64_t *A, int s, int in_level, int max_level, const sfg_geno *g, unsigned flags,
                                         int b0, int b1, int j0, int j1, int accumulate, uint64_t *acc) {
    ApiScope api_scope(ctx);
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    Shape sh = make_shape(g, flags);
    return matmul_accumulate(ctx, (const u64 *)A, s, in_level, max_level, sh, flags, b0, b1, j0, j1, accumulate, (u64 *)acc);
}
extern "C" int sfg_matmul_finalize_dev(sfg_ctx *ctx, const uint64_t *acc, int s, int max_level, int ncolb, int g0, int g1, int accumulate, uint64_t *out) {
    ApiScope api_scope(ctx);
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    return matmul_finalize(ctx, (const u64 *)acc, s, max_level, ncolb, ncolb, 0, g0, g1, nullptr, accumulate, (u64 *)out);
}

// the same with the accumulator slots of a giant-sharded rank: acc [ncolb][acc_giants][s][2][L][N] where slot g holds giant step
// giant_base + g (what a reduce-scatter over giant steps leaves on each rank); slots whose giant step is >= 91 are ignored
extern "C" int sfg_matmul_finalize_slots_dev(sfg_ctx *ctx, const uint64_t *acc, int s, int max_level, int ncolb, int acc_giants, int giant_base,
                                             int g0, int g1, int accumulate, uint64_t *out) {
    ApiScope api_scope(ctx);
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (acc_giants < 1) SFG_FAIL(ctx, "finalize: acc_giants must be positive");
    return matmul_finalize(ctx, (const u64 *)acc, s, max_level, ncolb, ncolb, 0, g0, g1, nullptr, accumulate, (u64 *)out, acc_giants, giant_base);
}

// SNP-block range [blk0, blk1) over the block columns of the STORED matrix: output columns for X, contraction rows for X^T
static int matmul_resident_range(sfg_ctx *ctx, const uint64_t *A, int s, int in_level, int max_level, const sfg_geno *g, unsigned flags,
                                 int blk0, int blk1, uint64_t *out, const double *rotf_ext, const I8RotPre *pre8 = nullptr) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    ctx->phases.clear();
    if (!rotf_ext && !pre8) ctx->i8_gen++;                               // a product that builds its own rot operands: whatever the int8 MAC had transposed is stale
    Shape sh = make_shape(g, flags);
    const int d = SFG_D, L = max_level, N = SFG_N;
    const size_t accw = (size_t)s * 2 * L * N;
    std::vector<uint8_t> giant_t; giant_table(sh, giant_t);
    int b0 = 0, b1 = sh.nbr, j0 = 0, j1 = sh.m_ct;
    if (sh.transposed) { b0 = blk0; b1 = blk1; } else { j0 = blk0; j1 = blk1; }
    if (b0 < 0 || b1 > sh.nbr || j0 < 0 || j1 > sh.m_ct || b0 > b1 || j0 > j1) SFG_FAIL(ctx, "matmul: SNP-block range out of bounds");
    const int m_out = j1 - j0;
    // column groups bounded by an accumulator budget (default 24 GiB)
    const size_t budget = ctx->cfg.acc_budget;
    int jg = (int)(budget / ((size_t)d * accw * 8)); if (jg < 1) jg = 1;
    // Several column groups would each rebuild the rotation cache of every block row (91 key switches per input
    // ciphertext).  When the whole cache fits (48 GiB; Q*X at 100k x 1M: 13 block rows = 28 GB) it is built once here.
    const double *rotf_all = rotf_ext, *rotsum_all = nullptr;
    if (rotf_ext && !mac_use_dma(ctx)) SFG_FAIL(ctx, "matmul: a prebuilt rotation cache needs the LDS-DMA MAC (the A/B build selected the register-staged kernel)");
    if (!rotf_ext && !pre8 && mac_use_dma(ctx) && j1 - j0 > jg && b1 > b0) {
        std::vector<int> plane_of, is_big; const int nplanes = mac_dma_planes(ctx, L, plane_of, is_big);
        if (nplanes < 0) return 1;
        const size_t rowf = (size_t)nplanes * N, per_row = (size_t)d * s * 2 * rowf;       // doubles per block row
        const int lev = in_level > max_level ? max_level : in_level, nl = lev + 1;
        if (nl >= L && ((size_t)(b1 - b0) * per_row + 3 * (size_t)s * 2 * rowf) * 8 <= (48ULL << 30)) {
            double *buf = nullptr; u64 *a_row = nullptr, *rotc = nullptr;
            const size_t ctw = (size_t)2 * nl * N;
            SFG_TRY(sfg_scratch(ctx, "mm.rotf", ((size_t)(b1 - b0) * per_row + 3 * (size_t)s * 2 * rowf) * 8, (void **)&buf));
            SFG_TRY(sfg_scratch(ctx, "mm.a_row", (size_t)s * ctw * 8, (void **)&a_row));
            SFG_TRY(sfg_scratch(ctx, "mm.rotc", 8, (void **)&rotc));                         // unused: the key switch writes the fp64 operand rows itself
            for (int bi = b0; bi < b1; bi++) SFG_TRY(build_rot_row(ctx, (const u64 *)A, s, in_level + 1, nl, lev, L, sh, bi, a_row, rotc, true, buf + (size_t)(bi - b0) * per_row));
            SFG_HIP(ctx, hipMemsetAsync(buf + (size_t)(b1 - b0) * per_row, 0, 3 * (size_t)s * 2 * rowf * 8, ctx->stream));   // k-slices read by a ragged last chunk
            rotf_all = buf;
            if (mac_dma_packed_mask(ctx, L)) {             // per MAC group (as matmul_accumulate forms them): sum of its k-slices
                const int G = std::min(ctx->cfg.mm_group, b1 - b0), ngrp = (b1 - b0 + G - 1) / G;
                double *rs = nullptr;
                SFG_TRY(sfg_scratch(ctx, "mm.rotsum_all", (size_t)ngrp * s * 2 * rowf * 8, (void **)&rs));
                for (int gi = 0; gi < ngrp; gi++) {
                    const int ng = std::min(G, b1 - b0 - gi * G);
                    SFG_TRY(launch_rot_sum(ctx, buf + (size_t)gi * G * per_row, (size_t)s * 2, ng * d, L, rs + (size_t)gi * s * 2 * rowf));
                }
                rotsum_all = rs;
            }
        }
    }
    // With the product-wide cache the accumulate passes do no key switching, so the giant-step alignment of pass k runs on the
    // auxiliary stream beside the encode + MAC of pass k+1 (two accumulator buffers of half the budget each).
    const bool overlap = (rotf_all || pre8) && !ctx->cfg.no_overlap;
    if (overlap) { jg = (jg + 1) / 2; }
    hipStream_t main_stream = ctx->stream;
    int k = 0;
    for (int ja = j0; ja < j1; ja += jg, k++) {
        const int jb = std::min(j1, ja + jg);
        u64 *acc = nullptr;
        SFG_TRY(sfg_scratch(ctx, "mm.acc", (size_t)jg * d * accw * 8 * (overlap ? 2 : 1), (void **)&acc));
        if (overlap) {
            acc += (size_t)(k & 1) * jg * d * accw;
            if (k >= 2) SFG_HIP(ctx, hipStreamWaitEvent(main_stream, ctx->ev_pipe[2 + (k & 1)], 0));    // pass k-2 has been aligned out of this buffer
        }
        int rc = matmul_accumulate(ctx, (const u64 *)A, s, in_level, max_level, sh, flags, b0, b1, ja, jb, 0, acc, rotf_all, rotsum_all, pre8);
        if (rc) return rc;
        if (overlap) {
            SFG_TRY(sfg_stream_after(ctx, ctx->aux_stream, main_stream));
            { AuxScope aux(ctx); rc = matmul_finalize(ctx, acc, s, max_level, jb - ja, m_out, ja - j0, 0, d, &giant_t, 0, (u64 *)out); }
            if (rc) return rc;
            SFG_HIP(ctx, hipEventRecord(ctx->ev_pipe[2 + (k & 1)], ctx->aux_stream));
        } else {
            rc = matmul_finalize(ctx, acc, s, max_level, jb - ja, m_out, ja - j0, 0, d, &giant_t, 0, (u64 *)out);
            if (rc) return rc;
        }
    }
    if (overlap) SFG_TRY(sfg_stream_after(ctx, main_stream, ctx->aux_stream));      // outputs are complete in main-stream order
    return 0;
}
extern "C" int sfg_matmul_resident_range_dev(sfg_ctx *ctx, const uint64_t *A, int s, int in_level, int max_level, const sfg_geno *g, unsigned flags,
                                             int blk0, int blk1, uint64_t *out) {
    ApiScope api_scope(ctx);
    return matmul_resident_range(ctx, A, s, in_level, max_level, g, flags, blk0, blk1, out, nullptr);
}

// ---------------------------------------------------------------- the rotation cache as an object (multi-GPU runs)
// The baby-step rotation cache of a product (matmult.go:1083-1119, 1373-1377: rotCache[i][baby] = RotateRight(A[i][bi], -baby)) in the MAC's fp64
// operand layout  cache[bi - row0][baby < 91][i < s][poly < 2][rowf]  (+ 3 zero k-slices).  Every rank of an output-sharded product (Q*X) needs the
// cache of ALL operand block rows; instead of rebuilding it on every rank, rank r key-switches the inputs (bi, i) of its job range
// (job = bi*s + i; the decomposition of an input is shared by its 91 rotations, so inputs - not baby steps - are the unit), the ranks all-gather
// the job-major staging buffers and scatter them into the MAC layout.  All 91 baby steps are rotated (a superset of the reference's active table
// whenever the operand has a ragged block only; rotations of inactive baby steps meet zero plaintexts), so the cache does not depend on a rank's window.
static int rotcache_rowf(sfg_ctx *ctx, int L, size_t &rowf) {
    if (!mac_use_dma(ctx)) SFG_FAIL(ctx, "rotation cache objects need the LDS-DMA MAC (the A/B build selected the register-staged kernel)");
    if (L < 1 || L > ctx->nq) SFG_FAIL(ctx, "rotcache: max_level out of range");
    std::vector<int> plane_of, is_big; const int nplanes = mac_dma_planes(ctx, L, plane_of, is_big);
    if (nplanes < 0) return 1;
    rowf = (size_t)nplanes * SFG_N; return 0;
}
extern "C" int sfg_rotcache_layout(sfg_ctx *ctx, int s, int max_level, size_t *job_doubles, size_t *tail_doubles) {
    size_t rowf = 0; SFG_TRY(rotcache_rowf(ctx, max_level, rowf));
    if (job_doubles) *job_doubles = (size_t)SFG_D * 2 * rowf;
    if (tail_doubles) *tail_doubles = (size_t)3 * s * 2 * rowf;
    return 0;
}
extern "C" int sfg_rotcache_build_jobs_dev(sfg_ctx *ctx, const uint64_t *A, int s, int in_level, int max_level, int nbr, int job0, int job1, double *staged) {
    ApiScope api_scope(ctx);
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    const int N = SFG_N, d = SFG_D, L = max_level;
    size_t rowf = 0; SFG_TRY(rotcache_rowf(ctx, L, rowf));
    const int lev = in_level > max_level ? max_level : in_level, nl = lev + 1, nl_in = in_level + 1;
    if (nl < L || in_level >= ctx->nq) SFG_FAIL(ctx, "rotcache: input level %d unusable with max_level = %d", in_level, max_level);
    if (s < 1 || nbr < 1 || job0 < 0 || job1 > nbr * s || job0 > job1) SFG_FAIL(ctx, "rotcache: job range out of bounds");
    const size_t ctw = (size_t)2 * nl * N;
    constexpr int CH = 64;                                   // inputs per key-switch call
    u64 *a_in = nullptr;
    SFG_TRY(sfg_scratch(ctx, "rc.in", (size_t)CH * ctw * 8, (void **)&a_in));
    PhaseTimer t(ctx, "rotate");
    for (int c0 = job0; c0 < job1; c0 += CH) {
        const int nj = std::min(CH, job1 - c0);
        for (int k = 0; k < nj; k++) {
            const int job = c0 + k, bi = job / s, i = job % s;
            const u64 *src = (const u64 *)A + ((size_t)i * nbr + bi) * 2 * nl_in * N;
            if (nl == nl_in) SFG_HIP(ctx, hipMemcpyAsync(a_in + (size_t)k * ctw, src, ctw * 8, hipMemcpyDeviceToDevice, ctx->stream));
            else hipLaunchKernelGGL(k_drop_level, dim3((unsigned)(2 * nl * (N / 256))), dim3(256), 0, ctx->stream, src, a_in + (size_t)k * ctw, nl_in, nl);
        }
        SFG_HIP(ctx, hipGetLastError());
        std::vector<int> nrv((size_t)d * nj), inv((size_t)d * nj); std::vector<size_t> slot((size_t)d * nj);
        for (int baby = 0; baby < d; baby++) for (int k = 0; k < nj; k++) {        // key-major job order, job-major output
            nrv[(size_t)baby * nj + k] = -baby; inv[(size_t)baby * nj + k] = k; slot[(size_t)baby * nj + k] = (size_t)(c0 - job0 + k) * d + baby;
        }
        SFG_TRY(launch_rotate_right_indexed_f64(ctx, a_in, nj, staged, d * nj, lev, nrv.data(), inv.data(), L, slot.data()));
    }
    t.stop(1);
    return 0;
}
extern "C" int sfg_rotcache_scatter_dev(sfg_ctx *ctx, const double *staged, int s, int max_level, int job0, int job1, int row0, int nrows, double *cache) {
    ApiScope api_scope(ctx);
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    ctx->i8_gen++;
    const int d = SFG_D;
    size_t rowf = 0; SFG_TRY(rotcache_rowf(ctx, max_level, rowf));
    if (s < 1 || nrows < 1 || job0 < row0 * s || job1 > (row0 + nrows) * s || job0 > job1) SFG_FAIL(ctx, "rotcache: job range outside the cache rows");
    const size_t jw = (size_t)2 * rowf;
    for (int job = job0; job < job1; job++) {
        const int bi = job / s, i = job % s;
        SFG_HIP(ctx, hipMemcpy2DAsync(cache + (((size_t)(bi - row0) * d) * s + i) * jw, (size_t)s * jw * 8, staged + (size_t)(job - job0) * d * jw, jw * 8, jw * 8, d,
                                      hipMemcpyDeviceToDevice, ctx->stream));
    }
    SFG_HIP(ctx, hipMemsetAsync(cache + (size_t)nrows * d * s * jw, 0, (size_t)3 * s * jw * 8, ctx->stream));   // k-slices a ragged last MAC chunk reads
    return 0;
}
// tabs (nullable): per block row of [b0, b1) the 91 active-baby flags (matmult.go:1326-1336; a superset is always right); null = all 91
// (every writer of a caller-held rotation cache bumps ctx->i8_gen: the int8 MAC keeps a transposed copy of a rot operand per (pointer, generation))
int rotcache_build_rows_tab(sfg_ctx *ctx, const u64 *A, int s, int in_level, int max_level, int nbr, int b0, int b1, const std::vector<std::vector<uint8_t>> *tabs, double *cache) {
    ctx->i8_gen++;
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    const int N = SFG_N, d = SFG_D, L = max_level;
    size_t rowf = 0; SFG_TRY(rotcache_rowf(ctx, L, rowf));
    const int lev = in_level > max_level ? max_level : in_level, nl = lev + 1;
    if (nl < L || in_level >= ctx->nq) SFG_FAIL(ctx, "rotcache: input level %d unusable with max_level = %d", in_level, max_level);
    if (s < 1 || b0 < 0 || b1 > nbr || b0 > b1) SFG_FAIL(ctx, "rotcache: block-row range out of bounds");
    const size_t ctw = (size_t)2 * nl * N, per_row = (size_t)d * s * 2 * rowf;
    u64 *a_row = nullptr, *rotc = nullptr;
    SFG_TRY(sfg_scratch(ctx, "mm.a_row", (size_t)s * ctw * 8, (void **)&a_row));
    SFG_TRY(sfg_scratch(ctx, "mm.rotc", 8, (void **)&rotc));
    const std::vector<uint8_t> all(d, 1);
    for (int bi = b0; bi < b1; bi++)
        SFG_TRY(build_rot_row_tab(ctx, A, s, in_level + 1, nl, lev, L, nbr, bi, tabs ? (*tabs)[bi - b0] : all, a_row, rotc, true, cache + (size_t)(bi - b0) * per_row));
    SFG_HIP(ctx, hipMemsetAsync(cache + (size_t)(b1 - b0) * per_row, 0, (size_t)3 * s * 2 * rowf * 8, ctx->stream));
    return 0;
}
extern "C" int sfg_rotcache_build_rows_dev(sfg_ctx *ctx, const uint64_t *A, int s, int in_level, int max_level, int nbr, int b0, int b1, double *cache) {
    ApiScope api_scope(ctx);
    return rotcache_build_rows_tab(ctx, (const u64 *)A, s, in_level, max_level, nbr, b0, b1, nullptr, cache);
}
// ---- the same cache held only as the int8 MAC's rot tiles (I8RotPre, kernels.hpp).  Built group by group: the fp64 operand rows of G block rows go through the
// product's own mm.rotf scratch, k_i8_pack_rot turns them into the group's tile buffers.  pre.G stays 0 (and nothing is held) when the context does not multiply
// every modulus on the int8 matrix core, the rows do not fit one MAC launch (2 s > 30: launch_mac_bc walks the rows 30 at a time) or the tiles do not fit the budget / the device.
// (the tile buffers are scratch entries "<prefix>.s<gi>" / "<prefix>.b<gi>" of the context: a scan's next call finds them in place - hipMalloc / hipFree of ~100 GB cost
//  0.2 to 3.3 s per call, measured - and they go back to the device like every kept buffer: under memory pressure from a later call, or sfg_ctx_release_scratch)
void i8_rotpre_free(I8RotPre &pre) { pre = I8RotPre(); }
int i8_rotpre_build(sfg_ctx *ctx, const u64 *A, int s, int in_level, int max_level, int nbr, const std::vector<std::vector<uint8_t>> *tabs, size_t budget_bytes, const char *prefix, I8RotPre &pre) {
    pre = I8RotPre();
    const int d = SFG_D, L = max_level;
    const auto &c = ctx->cfg;
    if (!c.assoc_i8 || !mac_use_dma(ctx) || !c.mac_bc || c.mac_plain_pt || !c.mac_i8 || !c.mac_i8_big || 2 * s > 30 || L < 1 || L > ctx->nq || nbr < 1) return 0;
    if (!mac_dma_packed_mask(ctx, L)) return 0;
    std::vector<int> plane_of, is_big; const int nplanes = mac_dma_planes(ctx, L, plane_of, is_big);
    if (nplanes < 0) { ctx->err.clear(); return 0; }
    // launch_mac_bc multiplies run by run of like moduli: one run of 35-bit moduli and at most one 46-bit modulus have one tile buffer each
    int l_big = -1, l_s0 = -1, n_s = 0, runs = 0;
    for (int l = 0; l < L; l++) {
        if (is_big[l]) { if (l_big >= 0) return 0; l_big = l; }
        else { if (l_s0 < 0) l_s0 = l; n_s++; if (l == 0 || is_big[l - 1]) runs++; }
    }
    if (runs != 1) return 0;
    const size_t rowf = (size_t)nplanes * SFG_N;
    auto tiles_of = [&](int Gc) { size_t t = 0; for (int b = 0; b < nbr; b += Gc) { const int ng = std::min(Gc, nbr - b); t += mac_i8_rot_tile_bytes(ng * d, n_s, 5) + (l_big >= 0 ? mac_i8_rot_tile_bytes(ng * d, 1, 6) : 0); } return t; };
    int G = std::min(c.mm_group, nbr);
    // 16 (12) block rows per MAC group where the HBM takes the larger plaintext panel (two of them: the encode of a launch runs beside the previous launch's
    // MAC) and plaintext tiles: half the launches, half the accumulator read-modify-writes (0.256 against 0.272 s per batch at 500 000 samples, 8 batches)
    if (c.mm_group_auto && nbr > G) {
        size_t fr = 0, tot = 0;
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) {
            size_t have = fr;
            for (const auto &kv : ctx->pool) if (kv.first.rfind(prefix, 0) == 0 || kv.first == "mm.pt" || kv.first == "mm.rotf" || kv.first.rfind("mi8.", 0) == 0) have += kv.second.second;
            for (int cand : {16, 12}) {
                const int G2 = std::min(cand, nbr);
                if (G2 <= G) break;
                const size_t panel = (size_t)G2 * d * d * L * (SFG_N / 2) * 8 * 2, rotf = ((size_t)G2 * d + 3) * s * 2 * rowf * 8;
                const size_t need = tiles_of(G2) + panel + rotf + mac_i8_stream_bytes(G2 * d, n_s, 5, 0) + (l_big >= 0 ? mac_i8_stream_bytes(G2 * d, 1, 6, 0) : 0) + (24ULL << 30);
                if (need <= have && tiles_of(G2) <= budget_bytes) { G = G2; break; }
            }
        }
    }
    if ((long long)G * d * 6 >= 131072 || tiles_of(G) > budget_bytes) return 0;
    // every buffer first (the fp64 rows of one group in the product's own mm.rotf, the tile buffers of all groups); where the larger groups do not fit after
    // all - another context on the device - the default group size is tried before the caller is told to fall back
    double *tmp = nullptr;
    auto alloc_all = [&](int Gc) -> bool {
        const int ngc = (nbr + Gc - 1) / Gc;
        pre.As.assign(ngc, nullptr); pre.Ab.assign(ngc, nullptr);
        int arc = sfg_scratch(ctx, "mm.rotf", ((size_t)Gc * d + 3) * s * 2 * rowf * 8, (void **)&tmp);
        for (int gi = 0; gi < ngc && !arc; gi++) {
            const int ng = std::min(Gc, nbr - gi * Gc);
            char nm[48];
            snprintf(nm, sizeof nm, "%s.s%d", prefix, gi);
            arc = sfg_scratch(ctx, nm, mac_i8_rot_tile_bytes(ng * d, n_s, 5), (void **)&pre.As[gi]);
            if (!arc && l_big >= 0) { snprintf(nm, sizeof nm, "%s.b%d", prefix, gi); arc = sfg_scratch(ctx, nm, mac_i8_rot_tile_bytes(ng * d, 1, 6), (void **)&pre.Ab[gi]); }
        }
        if (arc) { ctx->err.clear(); i8_rotpre_free(pre); }
        return !arc;
    };
    if (!alloc_all(G)) {
        const int G0 = std::min(c.mm_group, nbr);
        if (G0 == G) return 0;
        G = G0;
        if (!alloc_all(G)) return 0;
    }
    const int ngrp = (nbr + G - 1) / G;
    for (int gi = 0; gi < ngrp; gi++) {
        const int bg = gi * G, ng = std::min(G, nbr - bg);
        std::vector<std::vector<uint8_t>> sub;
        if (tabs) sub.assign(tabs->begin() + bg, tabs->begin() + bg + ng);
        int rc = rotcache_build_rows_tab(ctx, A, s, in_level, max_level, nbr, bg, bg + ng, tabs ? &sub : nullptr, tmp);
        if (!rc) rc = launch_i8_pack_rot_to(ctx, tmp, (size_t)s * 2 * rowf, rowf, plane_of[l_s0], ng * d, 2 * s, l_s0, n_s, false, pre.As[gi]);
        if (!rc && l_big >= 0) rc = launch_i8_pack_rot_to(ctx, tmp, (size_t)s * 2 * rowf, rowf, plane_of[l_big], ng * d, 2 * s, l_big, 1, true, pre.Ab[gi]);
        if (rc) { i8_rotpre_free(pre); return rc; }
    }
    pre.G = G; pre.nbr = nbr; pre.s = s; pre.L = L;
    return 0;
}
int matmul_resident_range_i8pre(sfg_ctx *ctx, const I8RotPre &pre, int s, int max_level, const sfg_geno *g, unsigned flags, int blk0, int blk1, uint64_t *out) {
    ApiScope api_scope(ctx);
    if (!pre.G) SFG_FAIL(ctx, "matmul: no int8 rot tiles");
    return matmul_resident_range(ctx, nullptr, s, max_level, max_level, g, flags, blk0, blk1, out, nullptr, &pre);
}
// accumulate phase of a product (sfg_matmul_accumulate_rc_dev) against int8 rot tiles that cover ALL operand block rows of the (possibly transposed) matrix: the
// multi-GPU engine's Q' X^T multiplies one output block column per call against the tiles of the rank's own block rows (mgpu.hip)
int matmul_accumulate_i8pre(sfg_ctx *ctx, const I8RotPre &pre, int s, int max_level, const sfg_geno *g, unsigned flags, int j0, int j1, int accumulate, uint64_t *acc, size_t acc_col_words) {
    ApiScope api_scope(ctx);
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (!pre.G) SFG_FAIL(ctx, "matmul: no int8 rot tiles");
    Shape sh = make_shape(g, flags);
    return matmul_accumulate(ctx, nullptr, s, max_level, max_level, sh, flags, 0, pre.nbr, j0, j1, accumulate, (u64 *)acc, nullptr, nullptr, &pre, acc_col_words);
}
// GetDiagBool (matmult.go:627-631) for other translation units
int sfg_diag_bool(int r, int c, int dim, int index) { return diag_bool(r, c, dim, index); }
// the products on a prebuilt cache that covers exactly the operand block rows the call contracts over ([0, nbr) for X, [blk0, blk1) / [b0, b1) for X^T)
extern "C" int sfg_matmul_resident_range_rc_dev(sfg_ctx *ctx, const double *cache, int s, int max_level, const sfg_geno *g, unsigned flags,
                                                int blk0, int blk1, uint64_t *out) {
    ApiScope api_scope(ctx);
    if (!cache) SFG_FAIL(ctx, "matmul: null rotation cache");
    return matmul_resident_range(ctx, nullptr, s, max_level, max_level, g, flags, blk0, blk1, out, cache);
}
extern "C" int sfg_matmul_accumulate_rc_dev(sfg_ctx *ctx, const double *cache, int s, int max_level, const sfg_geno *g, unsigned flags,
                                            int b0, int b1, int j0, int j1, int accumulate, uint64_t *acc) {
    ApiScope api_scope(ctx);
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (!cache) SFG_FAIL(ctx, "matmul: null rotation cache");
    if (!mac_use_dma(ctx)) SFG_FAIL(ctx, "matmul: a prebuilt rotation cache needs the LDS-DMA MAC (the A/B build selected the register-staged kernel)");
    Shape sh = make_shape(g, flags);
    return matmul_accumulate(ctx, nullptr, s, max_level, max_level, sh, flags, b0, b1, j0, j1, accumulate, (u64 *)acc, cache, nullptr);
}

extern "C" int sfg_matmul_resident_dev(sfg_ctx *ctx, const uint64_t *A, int s, int in_level, int max_level, const sfg_geno *g, unsigned flags, uint64_t *out) {
    ApiScope api_scope(ctx);
    const size_t nb = (g->ncol + SFG_SLOTS - 1) / SFG_SLOTS;    // SNP blocks = block columns of the stored matrix
    return sfg_matmul_resident_range_dev(ctx, A, s, in_level, max_level, g, flags, 0, (int)nb, out);
}

// MatMult4Stream(cps, A, gfs, maxLevel, computeSquaredSum, square, nproc) with host buffers (matmult.go:1238)
extern "C" int sfg_matmul_stream(sfg_ctx *ctx, const uint64_t *A_host, int s, int in_level, int max_level,
                                 const int8_t *geno_host, size_t nrow, size_t ncol, size_t ld, unsigned flags,
                                 uint64_t *out_host, double *sum_host, double *sqsum_host) {
    ApiScope api_scope(ctx);
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    sfg_geno *g = nullptr;
    SFG_TRY(sfg_geno_upload(ctx, geno_host, nrow, ncol, ld, &g));
    Shape sh = make_shape(g, flags);
    const size_t a_words = (size_t)s * sh.nbr * 2 * (in_level + 1) * SFG_N, o_words = (size_t)s * sh.m_ct * 2 * max_level * SFG_N;
    u64 *dA = nullptr, *dO = nullptr; int rc = 0;
    if (hipMalloc(&dA, a_words * 8) != hipSuccess || hipMalloc(&dO, o_words * 8) != hipSuccess) { rc = 1; ctx->err = "matmul_stream: out of device memory"; }
    if (!rc && hipMemcpy(dA, A_host, a_words * 8, hipMemcpyHostToDevice) != hipSuccess) { rc = 1; ctx->err = "matmul_stream: upload failed"; }
    if (!rc && (sum_host || sqsum_host)) rc = sfg_geno_colsums(ctx, g, sum_host, sqsum_host);   // sums are taken before squaring (:1297-1303)
    if (!rc) rc = sfg_matmul_resident_dev(ctx, (const uint64_t *)dA, s, in_level, max_level, g, flags, (uint64_t *)dO);
    if (!rc && (hipMemcpyAsync(out_host, dO, o_words * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess)) { rc = 1; ctx->err = "matmul_stream: download failed"; }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(dA); (void)hipFree(dO); sfg_geno_free(ctx, g);
    if (!rc) rc = sfg_encoder_check(ctx);
    return rc;
}


// ---------------------------------------------------------------- A9 with the reference's on-disk cache (A11 / F2)
// MatMult4StreamCompute reading DiagCache files written by MatMult4StreamPreprocess of a CPU party (matmult.go:1043-1236,
// filestream.go:19-282): <prefix>_<bi>.bin = header {vectorLen, level, scale bits, n, numModuli, rowSize} (6 x u64 LE) + d baby
// flags + d giant flags, then one record per active diagonal: u64 LE length, u32 LE shift, and per block column a u8 isEmpty flag
// followed (when not empty) by numModuli x n coefficients as big-endian u64 (ring.WriteCoeffsTo) in NTT + Montgomery form.
// A file is streamed once per column pass, one giant step (<= 91 records) at a time: the records of giant g form the plaintext panel
// pt[j][baby] of ONE MAC launch with columns = block columns j and K = 91 baby steps, accumulated into acc[j][g].
struct DiagCacheHdr { uint64_t vectorLen, level, n, numModuli, rowSize; double scale; std::vector<uint8_t> baby, giant; long data_pos; };
static int dc_open(sfg_ctx *ctx, const std::string &fn, FILE **fp, DiagCacheHdr &h) {
    FILE *f = fopen(fn.c_str(), "rb");
    if (!f) SFG_FAIL(ctx, "matmul_from_cache: cannot open %s", fn.c_str());          // os.Open panics in the reference (filestream.go:59-61)
    unsigned char b[48];
    auto le = [&](int k) { uint64_t v = 0; for (int i = 0; i < 8; i++) v |= (uint64_t)b[8 * k + i] << (8 * i); return v; };
    h.baby.assign(SFG_D, 0); h.giant.assign(SFG_D, 0);
    if (fread(b, 1, 48, f) != 48 || fread(h.baby.data(), 1, SFG_D, f) != (size_t)SFG_D || fread(h.giant.data(), 1, SFG_D, f) != (size_t)SFG_D) { fclose(f); SFG_FAIL(ctx, "matmul_from_cache: short header in %s", fn.c_str()); }
    h.vectorLen = le(0); h.level = le(1); uint64_t sb = le(2); memcpy(&h.scale, &sb, 8); h.n = le(3); h.numModuli = le(4); h.rowSize = le(5);
    h.data_pos = ftell(f);
    *fp = f; return 0;
}
// raw: [91 babies][jp][L][N] big-endian Montgomery words, present[baby*jp + j] != 0 where a plaintext was read.
// panel[(j*91 + baby)][l][N/2] = canonical (or packed-limb) half row; asym counts words whose mirror differs (not a real-slot plaintext)
__global__ void __launch_bounds__(256) k_cache_to_panel(const u64 *raw, const uint8_t *present, u64 *panel, int jp, int L, unsigned packed_mask,
                                                        const u64 *r64inv, const ModConst *modc, unsigned long long *asym) {
    const int N = SFG_N, n = N / 2;
    const size_t row = blockIdx.x / (n / 256);                 // over [baby][j][l]
    const int l = (int)(row % L); const size_t bj = row / L; const int j = (int)(bj % jp), baby = (int)(bj / jp);
    const int x = (int)(blockIdx.x % (n / 256)) * 256 + threadIdx.x;
    u64 *dst = panel + (((size_t)j * SFG_D + baby) * L + l) * n;
    const bool packed = (packed_mask >> l) & 1u;
    if (!present[bj]) { dst[x] = packed ? PACKED_ZERO : 0ULL; return; }
    const u64 *src = raw + row * N;
    const u64 be = src[x], bm = src[N - 1 - x];
    if (be != bm) atomicAdd(asym, 1ULL);
    const u64 w = __builtin_bswap64(be);
    const u64 v = d_mulmod_u64(w % modc[l].qi, r64inv[l], modc[l].qi);          // ring.InvMForm: x * 2^-64 mod q
    dst[x] = packed ? pack_limbs(v) : v;
}

// ---- MatMult4StreamPreprocess with its reference-format output (matmult.go:914-1041 -> filestream.go:144-231): the DiagCache files of a resident matrix,
// written from DEVICE-encoded diagonals, so that a CPU-only party (or sfg_matmul_from_cache) can multiply from them.
// canonical NTT words pt[nshift][LV][N] -> payload words of ring.WriteCoeffsTo: MForm (x 2^64 mod q, matmult.go:401-440) then big-endian
__global__ void __launch_bounds__(256) k_cache_payload(const u64 *pt, u64 *out, int LV, const u64 *r64, const ModConst *modc) {
    const int N = SFG_N; const size_t row = blockIdx.x / (N / 256); const int l = (int)(row % LV);
    const size_t x = row * N + (blockIdx.x % (N / 256)) * 256 + threadIdx.x;
    out[x] = __builtin_bswap64(d_mulmod_u64(pt[x], r64[l], modc[l].qi));
}
extern "C" int sfg_diagcache_write(sfg_ctx *ctx, const sfg_geno *g, unsigned flags, int max_level, const char *prefix, int *files_written) {
    ApiScope api_scope(ctx);
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (files_written) *files_written = 0;
    if (!g || !prefix) SFG_FAIL(ctx, "sfg_diagcache_write: null argument");
    if (flags & ~SFG_TRANSPOSE) SFG_FAIL(ctx, "sfg_diagcache_write: only the transpose flag is meaningful here (MatMult4StreamPreprocess neither squares nor sums)");
    const int LV = max_level + 1;                    // EncodeDiagWithEncoder makes level-maxLevel plaintexts: maxLevel + 1 moduli rows are written, maxLevel are multiplied
    if (max_level < 1 || LV > ctx->nq) SFG_FAIL(ctx, "sfg_diagcache_write: max_level out of range");
    if (g->packed) {
        sfg_geno *u = nullptr; SFG_TRY(sfg_geno_unpack(ctx, g, &u));
        const int rc = sfg_diagcache_write(ctx, u, flags, max_level, prefix, files_written); sfg_geno_free(ctx, u); return rc;
    }
    const Shape sh = make_shape(g, flags);
    const int N = SFG_N, d = SFG_D, slots = SFG_SLOTS;
    const size_t ptw = (size_t)LV * N;                                   // words per plaintext
    const size_t row_size = 4 + (1 + ptw * 8) * (size_t)sh.m_ct;         // filestream.go:166 rowSize
    // 2^64 mod q_l per modulus (ring.MForm's constant)
    u64 r64_h[SFG_MAXMOD];
    for (int l = 0; l < LV; l++) { const unsigned __int128 t = ((unsigned __int128)1 << 64) % ctx->q[l]; r64_h[l] = (u64)t; }
    u64 *r64_d = nullptr; int8_t *skew = nullptr; u64 *half = nullptr, *full = nullptr, *pay_d = nullptr; unsigned char *pay_h = nullptr;
    const int BATCH = 64;                                                // shifts per device batch: 64 x LV x N x 8 = 50 MB of payload per block column
    SFG_TRY(sfg_scratch(ctx, "dcw.r64", SFG_MAXMOD * 8, (void **)&r64_d));
    SFG_TRY(sfg_scratch(ctx, "enc.skew", (size_t)slots * slots, (void **)&skew));
    SFG_TRY(sfg_scratch(ctx, "enc.half", (size_t)BATCH * LV * (N / 2) * 8, (void **)&half));
    SFG_TRY(sfg_scratch(ctx, "dcw.full", (size_t)BATCH * ptw * 8, (void **)&full));
    SFG_TRY(sfg_scratch(ctx, "dcw.pay", (size_t)BATCH * ptw * 8, (void **)&pay_d));
    SFG_HIP(ctx, hipMemcpyAsync(r64_d, r64_h, SFG_MAXMOD * 8, hipMemcpyHostToDevice, ctx->stream));
    SFG_HIP(ctx, hipHostMalloc((void **)&pay_h, (size_t)BATCH * ptw * 8, hipHostMallocDefault));
    int rc = 0, written = 0;
    for (int bi = 0; bi < sh.nbr && !rc; bi++) {
        const std::string fn = std::string(prefix) + "_" + std::to_string(bi) + ".bin";
        if (FILE *t = fopen(fn.c_str(), "rb")) { fclose(t); continue; }             // NewDiagCacheStream(..., isWrite): an existing file is kept (filestream.go:48-54, matmult.go:928-931)
        const int nr = sh.rows_of(bi);
        // active shifts and the baby / giant tables of the header (matmult.go:962-972): union over the block columns
        std::vector<uint8_t> baby_t(d, 0), giant_t(d, 0), shift_t(slots, 0);
        std::vector<std::vector<uint8_t>> has(sh.m_ct, std::vector<uint8_t>(slots, 0));
        for (int shift = 0; shift < slots; shift++) {
            bool any = false;
            for (int bj = 0; bj < sh.m_ct; bj++) { has[bj][shift] = diag_bool(nr, sh.cols_of(bj), slots, -shift) ? 1 : 0; any = any || has[bj][shift]; }
            if (any) { baby_t[shift % d] = 1; giant_t[shift / d] = 1; shift_t[shift] = 1; }
        }
        const std::string tmp = fn + ".part";
        FILE *f = fopen(tmp.c_str(), "wb");
        if (!f) { ctx->err = "sfg_diagcache_write: cannot create " + tmp; rc = 1; break; }
        {   // header: 6 x u64 LE {vectorLen, level, scale bits, n, numModuli, rowSize}, then d baby flags, d giant flags (filestream.go:154-187)
            uint64_t sb; const double sc = ctx->scale; memcpy(&sb, &sc, 8);
            const uint64_t hdr[6] = {(uint64_t)sh.m_ct, (uint64_t)max_level, sb, (uint64_t)N, (uint64_t)LV, (uint64_t)row_size};
            unsigned char hb[48]; for (int k = 0; k < 6; k++) for (int i = 0; i < 8; i++) hb[8 * k + i] = (unsigned char)(hdr[k] >> (8 * i));
            if (fwrite(hb, 1, 48, f) != 48 || fwrite(baby_t.data(), 1, d, f) != (size_t)d || fwrite(giant_t.data(), 1, d, f) != (size_t)d) rc = 1;
        }
        // one file position per record: records are laid out in increasing shift (the order of the reference's job feeder, matmult.go:989-999), a record's
        // plaintext j sits at a known offset, so the block columns can be encoded one after the other (one skew per block) and written in place
        std::vector<long long> rec_pos(slots, -1); long long pos = 48 + 2 * d;
        for (int shift = 0; shift < slots && !rc; shift++) if (shift_t[shift]) {
            size_t len = 4; for (int bj = 0; bj < sh.m_ct; bj++) len += 1 + (has[bj][shift] ? ptw * 8 : 0);
            unsigned char head[12]; for (int i = 0; i < 8; i++) head[i] = (unsigned char)((uint64_t)len >> (8 * i));
            for (int i = 0; i < 4; i++) head[8 + i] = (unsigned char)((uint32_t)shift >> (8 * i));
            if (fseeko(f, pos, SEEK_SET) || fwrite(head, 1, 12, f) != 12) { rc = 1; break; }
            rec_pos[shift] = pos + 12; pos += 8 + (long long)len;
        }
        for (int bj = 0; bj < sh.m_ct && !rc; bj++) {
            const int nc = sh.cols_of(bj);
            rc = launch_skew(ctx, sh.block(bi, bj), sh.ld, nr, nc, sh.transposed ? 1 : 0, 0, skew);
            for (int s0 = 0; s0 < slots && !rc; s0 += BATCH) {
                int lo = -1, hi = -1;
                for (int sft = s0; sft < std::min(slots, s0 + BATCH); sft++) if (has[bj][sft]) { if (lo < 0) lo = sft; hi = sft + 1; }
                if (lo >= 0) {
                    const int nb = hi - lo;
                    rc = launch_encode_rows(ctx, skew, lo, nb, LV, half, true);
                    if (!rc) rc = launch_expand_half(ctx, half, full, (size_t)nb * LV);
                    if (!rc) {
                        hipLaunchKernelGGL(k_cache_payload, dim3((unsigned)((size_t)nb * LV * (N / 256))), dim3(256), 0, ctx->stream, full, pay_d, LV, r64_d, ctx->modc);
                        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(pay_h, pay_d, (size_t)nb * ptw * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                            hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "sfg_diagcache_write: device work failed"; rc = 1; }
                    }
                }
                for (int sft = s0; sft < std::min(slots, s0 + BATCH) && !rc; sft++) if (shift_t[sft]) {
                    // plaintext bj of record sft: isEmpty byte, then the payload
                    long long off = rec_pos[sft]; for (int k = 0; k < bj; k++) off += 1 + (has[k][sft] ? (long long)ptw * 8 : 0);
                    const unsigned char flag = has[bj][sft] ? 0 : 1;
                    if (fseeko(f, off, SEEK_SET) || fwrite(&flag, 1, 1, f) != 1) { rc = 1; break; }
                    if (has[bj][sft] && fwrite(pay_h + (size_t)(sft - lo) * ptw * 8, 1, ptw * 8, f) != ptw * 8) { rc = 1; break; }
                }
            }
        }
        if (fclose(f)) rc = 1;
        if (!rc && rename(tmp.c_str(), fn.c_str())) rc = 1;
        if (rc) { remove(tmp.c_str()); if (ctx->err.empty()) ctx->err = "sfg_diagcache_write: I/O error on " + fn; break; }
        written++;
    }
    (void)hipHostFree(pay_h);
    if (files_written) *files_written = written;
    if (!rc) rc = sfg_encoder_check(ctx);
    return rc;
}

extern "C" int sfg_diagcache_header(sfg_ctx *ctx, const char *prefix, int block_row, uint64_t hdr[6]) {
    FILE *f = nullptr; DiagCacheHdr h;
    SFG_TRY(dc_open(ctx, std::string(prefix) + "_" + std::to_string(block_row) + ".bin", &f, h));
    fclose(f);
    uint64_t sb; memcpy(&sb, &h.scale, 8);
    hdr[0] = h.vectorLen; hdr[1] = h.level; hdr[2] = sb; hdr[3] = h.n; hdr[4] = h.numModuli; hdr[5] = h.rowSize;
    return 0;
}

extern "C" int sfg_matmul_from_cache(sfg_ctx *ctx, const uint64_t *A, int s, int in_level, int max_level, const char *prefix, int nbr, uint64_t *out) {
    ApiScope api_scope(ctx);
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    ctx->phases.clear();
    if (!mac_use_dma(ctx)) SFG_FAIL(ctx, "matmul_from_cache needs the LDS-DMA MAC (the A/B build selected the register-staged kernel)");
    const int N = SFG_N, d = SFG_D, L = max_level;
    if (L < 1 || L > ctx->nq) SFG_FAIL(ctx, "matmul_from_cache: max_level out of range");
    const int lev = in_level > max_level ? max_level : in_level, nl = lev + 1, nl_in = in_level + 1;
    if (nl < L) SFG_FAIL(ctx, "matmul_from_cache: input level %d has fewer than max_level = %d moduli", in_level, max_level);
    if (nbr < 1) SFG_FAIL(ctx, "matmul_from_cache: no block rows");
    // headers of every block row first: shapes must agree, giant tables are united (matmult.go:1074-1078)
    std::vector<DiagCacheHdr> hdrs(nbr); std::vector<FILE *> files(nbr, nullptr);
    auto close_all = [&]() { for (FILE *f : files) if (f) fclose(f); };
    for (int bi = 0; bi < nbr; bi++) {
        if (dc_open(ctx, std::string(prefix) + "_" + std::to_string(bi) + ".bin", &files[bi], hdrs[bi])) { close_all(); return 1; }
        const DiagCacheHdr &h = hdrs[bi];
        // a corrupt or hostile header must not size an allocation: moduli bounded by the ring, block columns by the largest matrix the slots can index,
        // the record size by the formula AND by the file itself
        long fsize = -1; { const long here = ftell(files[bi]); if (!fseek(files[bi], 0, SEEK_END)) fsize = ftell(files[bi]); fseek(files[bi], here, SEEK_SET); }
        if (h.n != (uint64_t)N || h.numModuli < (uint64_t)L || h.numModuli > (uint64_t)ctx->nmod || h.vectorLen != hdrs[0].vectorLen || h.vectorLen < 1 ||
            h.vectorLen > (1u << 20) || h.rowSize != 4 + (1 + h.n * h.numModuli * 8) * h.vectorLen || fsize < 0 || h.rowSize > (uint64_t)fsize) {
            close_all(); SFG_FAIL(ctx, "matmul_from_cache: %s_%d.bin does not match this ring (n = %llu, numModuli = %llu, vectorLen = %llu)", prefix, bi,
                                  (unsigned long long)h.n, (unsigned long long)h.numModuli, (unsigned long long)h.vectorLen);
        }
    }
    const int m_ct = (int)hdrs[0].vectorLen;
    std::vector<uint8_t> giant_t(d, 0);
    for (auto &h : hdrs) for (int g = 0; g < d; g++) giant_t[g] |= h.giant[g];
    const size_t accw = (size_t)s * 2 * L * N, ctw = (size_t)2 * nl * N, prow = (size_t)N / 2, plw = (size_t)L * prow;
    std::vector<int> plane_of, is_big; const int nplanes = mac_dma_planes(ctx, L, plane_of, is_big);
    if (nplanes < 0) { close_all(); return 1; }
    const size_t rowf = (size_t)nplanes * N;
    const unsigned packed_mask = mac_dma_packed_mask(ctx, L);
    int jp = (int)(ctx->cfg.acc_budget / ((size_t)d * accw * 8)); if (jp < 1) jp = 1; if (jp > m_ct) jp = m_ct;
    u64 *a_row = nullptr, *rotc = nullptr, *acc = nullptr, *panel = nullptr, *raw_d = nullptr, *inv_d = nullptr; double *rotf = nullptr, *rotsum = nullptr;
    uint8_t *present_d = nullptr; unsigned long long *asym_d = nullptr; u64 *raw_h = nullptr;
    int rc = 0;
    auto bail = [&](int r) { close_all(); if (raw_h) (void)hipHostFree(raw_h); return r; };
    if (sfg_scratch(ctx, "mm.a_row", (size_t)s * ctw * 8, (void **)&a_row) || sfg_scratch(ctx, "mm.rotc", 8, (void **)&rotc) ||
        sfg_scratch(ctx, "mm.rotf", ((size_t)d + 3) * s * 2 * rowf * 8, (void **)&rotf) || sfg_scratch(ctx, "mm.rotsum", (size_t)2 * s * 2 * rowf * 8, (void **)&rotsum) ||
        sfg_scratch(ctx, "mm.acc", (size_t)jp * d * accw * 8, (void **)&acc) || sfg_scratch(ctx, "dc.panel", (size_t)jp * d * plw * 8, (void **)&panel) ||
        sfg_scratch(ctx, "dc.raw", (size_t)d * jp * L * N * 8 + (size_t)d * jp + 64 + 8 * SFG_MAXMOD, (void **)&raw_d)) return bail(1);
    present_d = (uint8_t *)(raw_d + (size_t)d * jp * L * N);
    asym_d = (unsigned long long *)(present_d + (((size_t)d * jp + 63) & ~(size_t)63)); inv_d = (u64 *)(asym_d + 1);
    if (hipHostMalloc((void **)&raw_h, (size_t)d * jp * L * N * 8 + (size_t)d * jp, hipHostMallocDefault) != hipSuccess) { ctx->err = "matmul_from_cache: hipHostMalloc failed"; return bail(1); }
    uint8_t *present_h = (uint8_t *)(raw_h + (size_t)d * jp * L * N);
    {
        u64 inv[SFG_MAXMOD] = {0};
        for (int l = 0; l < L; l++) { const u64 q = ctx->q[l]; inv[l] = h_invmod((u64)((((u128)1) << 64) % q), q); }
        if (hipMemcpyAsync(inv_d, inv, sizeof inv, hipMemcpyHostToDevice, ctx->stream) != hipSuccess || hipMemsetAsync(asym_d, 0, 8, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "matmul_from_cache: setup copy failed"; return bail(1); }
    }
    std::vector<unsigned char> rec;
    for (int ja = 0; ja < m_ct && !rc; ja += jp) {
        const int jn = std::min(jp, m_ct - ja);
        if (hipMemsetAsync(acc, 0, (size_t)jn * d * accw * 8, ctx->stream) != hipSuccess) { ctx->err = "matmul_from_cache: memset failed"; rc = 1; break; }
        for (int bi = 0; bi < nbr && !rc; bi++) {
            const DiagCacheHdr &h = hdrs[bi]; FILE *f = files[bi];
            fseek(f, h.data_pos, SEEK_SET);
            try { rec.resize(h.rowSize); } catch (const std::exception &) { ctx->err = "matmul_from_cache: out of host memory for a record"; rc = 1; break; }
            rc = build_rot_row_tab(ctx, (const u64 *)A, s, nl_in, nl, lev, L, nbr, bi, h.baby, a_row, rotc, true, rotf);       // matmult.go:1083-1119
            if (!rc && hipMemsetAsync(rotf + (size_t)d * s * 2 * rowf, 0, (size_t)3 * s * 2 * rowf * 8, ctx->stream) != hipSuccess) rc = 1;
            if (!rc && packed_mask) rc = launch_rot_sum(ctx, rotf, (size_t)s * 2, d, L, rotsum);
            int cur_giant = -1; bool any = false;
            auto flush = [&]() -> int {                      // the buffered records of giant cur_giant -> panel -> one MAC launch
                if (!any) return 0;
                SFG_HIP(ctx, hipMemcpyAsync(raw_d, raw_h, (size_t)d * jn * L * N * 8, hipMemcpyHostToDevice, ctx->stream));
                SFG_HIP(ctx, hipMemcpyAsync(present_d, present_h, (size_t)d * jn, hipMemcpyHostToDevice, ctx->stream));
                hipLaunchKernelGGL(k_cache_to_panel, dim3((unsigned)((size_t)d * jn * L * (N / 2 / 256))), dim3(256), 0, ctx->stream, raw_d, present_d, panel, jn, L, packed_mask,
                                   inv_d, ctx->modc, asym_d);
                SFG_HIP(ctx, hipGetLastError());
                MacStrides st;
                st.rot_k = (size_t)s * ctw; st.rot_r = (size_t)nl * N;
                st.pt_k = plw; st.pt_n = (size_t)d * plw; st.pt_half = true; st.pt_packed = packed_mask != 0; st.i8 = ctx->cfg.mac_i8 && packed_mask; st.i8_big = st.i8 && ctx->cfg.mac_i8_big;     // panel[j][baby]
                st.out_n = (size_t)d * accw; st.out_r = (size_t)L * N;                                           // acc[j][giant][r], column n = j
                PhaseTimer t(ctx, "mac");
                int r2 = launch_mac_dma(ctx, rotf, (size_t)s * 2, panel, acc + (size_t)cur_giant * accw, d, 2 * s, jn, L, 1, st, rotsum);
                t.stop(1);
                SFG_HIP(ctx, hipStreamSynchronize(ctx->stream));               // raw_h is refilled next
                any = false;
                return r2;
            };
            while (!rc) {                                   // ReadDiag (filestream.go:247-282) until EOF
                unsigned char l8[8];
                if (fread(l8, 1, 8, f) != 8) break;
                uint64_t len = 0; for (int i = 0; i < 8; i++) len |= (uint64_t)l8[i] << (8 * i);
                if (len > h.rowSize || len < 4 || fread(rec.data(), 1, len, f) != len) { ctx->err = "matmul_from_cache: truncated record"; rc = 1; break; }
                const int shift = (int)((uint32_t)rec[0] | (uint32_t)rec[1] << 8 | (uint32_t)rec[2] << 16 | (uint32_t)rec[3] << 24);
                if (shift < 0 || shift >= SFG_SLOTS) { ctx->err = "matmul_from_cache: shift out of range"; rc = 1; break; }
                const int giant = shift / d, baby = shift % d;
                if (giant != cur_giant) { rc = flush(); if (rc) break; cur_giant = giant; memset(present_h, 0, (size_t)d * jn); }
                size_t ptr = 4; const size_t plain_bytes = (size_t)h.n * h.numModuli * 8;
                for (uint64_t j = 0; j < h.vectorLen; j++) {
                    if (ptr >= len) { ctx->err = "matmul_from_cache: malformed record"; rc = 1; break; }
                    const bool empty = rec[ptr++] == 1;
                    if (empty) continue;
                    if (ptr + plain_bytes > len) { ctx->err = "matmul_from_cache: malformed record"; rc = 1; break; }
                    if ((int)j >= ja && (int)j < ja + jn) {
                        memcpy(raw_h + ((size_t)baby * jn + (j - ja)) * L * N, rec.data() + ptr, (size_t)L * N * 8);      // the first L of numModuli rows
                        present_h[(size_t)baby * jn + (j - ja)] = 1; any = true;
                    }
                    ptr += plain_bytes;
                }
            }
            if (!rc) rc = flush();
        }
        if (!rc) rc = matmul_finalize(ctx, acc, s, max_level, jn, m_ct, ja, 0, d, &giant_t, 0, (u64 *)out);
        if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) rc = 1;
    }
    unsigned long long asym = 0;
    if (!rc && hipMemcpy(&asym, asym_d, 8, hipMemcpyDeviceToHost) != hipSuccess) rc = 1;
    if (!rc && asym) { ctx->err = "matmul_from_cache: cached plaintexts are not mirror-symmetric (not encodings of real slot vectors)"; rc = 1; }
    return bail(rc);
}
