// genoio.hip — genotype input pipeline on the device (SURVEY §8b "Python callers/harness", §8f-3):
//   scripts/plinkBedToBinary.py  PLINK .bed (SNP-major 2-bit codes) -> sample-major int8 {00->2, 01->-1, 10->1, 11->0}
//   scripts/filterMatrix.py      keep rows / columns by byte filters           (fused into the decode)
//   scripts/transposeMatrix.py   [nrows x ncols] -> [ncols x nrows]
//   scripts/mergeMatrices.py     column-wise concatenation
// The packed image crosses PCIe (4x less than int8) and is expanded in HBM; results are resident sfg_geno handles.
#include "common.hpp"
#include "kernels.hpp"
#include <algorithm>

// tile = 64 SNPs x 64 bytes (256 samples). grid (ceil(bps/64), ceil(num_snp/64)), 256 threads.  lut: int8 value of 2-bit code c in byte c
// (.bed: 00 -> 2, 01 -> -1, 10 -> 1, 11 -> 0 = BED_LUT; .pgen genotype codes: c -> c, 3 -> -1)
constexpr unsigned BED_LUT = 0x0001FF02u;
__global__ void __launch_bounds__(256) k_bed_decode(const uint8_t *bed, size_t bps, size_t num_sample, size_t num_snp,
                                                    const int32_t *row_map, const int32_t *col_map, int8_t *out, size_t ld, unsigned lut) {
    __shared__ int8_t tile[256][65];                          // [sample][snp], padded
    const size_t b0 = (size_t)blockIdx.x * 64, j0 = (size_t)blockIdx.y * 64;
    const int t = threadIdx.x;
#pragma unroll 4
    for (int it = 0; it < 16; it++) {                          // coalesced along the bytes of one SNP
        const int idx = it * 256 + t, snp = idx >> 6, byte = idx & 63;
        uint32_t v = 0;
        if (j0 + snp < num_snp && b0 + byte < bps) v = bed[(j0 + snp) * bps + b0 + byte];
#pragma unroll
        for (int k = 0; k < 4; k++) tile[byte * 4 + k][snp] = (int8_t)(lut >> (8 * ((v >> (2 * k)) & 3)));
    }
    __syncthreads();
    for (int it = 0; it < 64; it++) {                          // 64 consecutive SNPs of one sample per wave
        const int idx = it * 256 + t, s = idx >> 6, snp = idx & 63;
        const size_t gs = b0 * 4 + s, gj = j0 + snp;
        if (gs < num_sample && gj < num_snp) {
            const long r = row_map ? row_map[gs] : (long)gs, c = col_map ? col_map[gj] : (long)gj;
            if (r >= 0 && c >= 0) out[(size_t)r * ld + (size_t)c] = tile[s][snp];
        }
    }
}
// 64 x 64 tiles. grid (ceil(ncol/64), ceil(nrow/64)), 256 threads
__global__ void __launch_bounds__(256) k_geno_transpose(const int8_t *in, size_t nrow, size_t ncol, size_t ld, int8_t *out) {
    __shared__ int8_t tile[64][65];
    const size_t c0 = (size_t)blockIdx.x * 64, r0 = (size_t)blockIdx.y * 64;
    const int t = threadIdx.x;
    for (int it = 0; it < 16; it++) { const int idx = it * 256 + t, r = idx >> 6, c = idx & 63; if (r0 + r < nrow && c0 + c < ncol) tile[r][c] = in[(r0 + r) * ld + c0 + c]; }
    __syncthreads();
    for (int it = 0; it < 16; it++) { const int idx = it * 256 + t, c = idx >> 6, r = idx & 63; if (r0 + r < nrow && c0 + c < ncol) out[(c0 + c) * nrow + r0 + r] = tile[r][c]; }
}

// decode + filter + transpose of a packed SNP range already in HBM, on the given queue (stream.hip streams batches through this)
int launch_bed_decode_lut(sfg_ctx *ctx, hipStream_t st, const uint8_t *dbed, size_t bps, size_t num_sample, size_t num_snp, const int32_t *rmap, const int32_t *cmap,
                          int8_t *out, size_t ld, unsigned lut) {
    hipLaunchKernelGGL(k_bed_decode, dim3((unsigned)((bps + 63) / 64), (unsigned)((num_snp + 63) / 64)), dim3(256), 0, st, dbed, bps, num_sample, num_snp, rmap, cmap, out, ld, lut);
    SFG_HIP(ctx, hipGetLastError());
    return 0;
}
int launch_bed_decode(sfg_ctx *ctx, hipStream_t st, const uint8_t *dbed, size_t bps, size_t num_sample, size_t num_snp, const int32_t *rmap, const int32_t *cmap,
                      int8_t *out, size_t ld) {
    return launch_bed_decode_lut(ctx, st, dbed, bps, num_sample, num_snp, rmap, cmap, out, ld, BED_LUT);
}
// ---- 2-bit packed residency (100k x 1M is 25 GB instead of 100 GB; 500k x 10M fits 8 GPUs).  Codes 0, 1, 2 = the genotype,
// 3 = missing; 4 consecutive columns per byte, low bits first; a row is ceil(ncol / 16) dwords.  The products expand one 8192 x 8192 block at a time
// into the int8 staging block the skew kernel reads (16 MB in, 64 MB out per block: ~1 % of a block's encode + MAC time).
// grid (ceil(ldb/4 / 256), nrow): one dword = 16 columns per thread
__global__ void __launch_bounds__(256) k_geno_pack(const int8_t *in, size_t ncol, size_t ld, unsigned *out, size_t ldw, unsigned *bad) {
    const size_t w = (size_t)blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (w >= ldw) return;
    unsigned v = 0, nb = 0;
    for (int k = 0; k < 16; k++) {
        const size_t c = w * 16 + k;
        int g = c < ncol ? in[r * ld + c] : 0;
        if (g > 2) nb++;
        v |= (unsigned)(g < 0 ? 3 : (g & 3)) << (2 * k);
    }
    out[r * ldw + w] = v;
    if (nb) atomicAdd(bad, nb);
}
// four 2-bit codes of one packed byte -> four int8 {0, 1, 2, -1} (one dword), without a table: spread the bit pairs to the bytes, then turn every 3 into 0xFF
__device__ __forceinline__ unsigned unpack4(unsigned x) {
    const unsigned t = (x | (x << 6) | (x << 12) | (x << 18)) & 0x03030303u;
    const unsigned m = (t + 0x7D7D7D7Du) & 0x80808080u;                   // bit 7 where the code is 3
    return t | ((m << 1) - (m >> 7));
}
// grid (ceil(nc / 4096), nr): thread = one packed dword = 16 columns, written as one 16-byte store (VEC: dword-aligned source, 16-byte aligned rows of out);
// the byte-wise form (thread = 4 columns) covers unaligned windows
template <bool VEC>
__global__ void __launch_bounds__(256) k_geno_unpack(const uint8_t *in, size_t ldb, size_t r0, size_t b0, size_t nc, int8_t *out, size_t ld_out) {
    const size_t r = blockIdx.y;
    if (VEC) {
        const size_t w = (size_t)blockIdx.x * 256 + threadIdx.x;           // dword index inside the window
        if (w * 16 >= nc) return;
        const unsigned v = *reinterpret_cast<const unsigned *>(in + (r0 + r) * ldb + b0 + w * 4);
        const uint4 o = make_uint4(unpack4(v & 0xFFu), unpack4((v >> 8) & 0xFFu), unpack4((v >> 16) & 0xFFu), unpack4(v >> 24));
        int8_t *dst = out + r * ld_out + w * 16;
        if (w * 16 + 16 <= nc) *reinterpret_cast<uint4 *>(dst) = o;
        else { const unsigned ov[4] = {o.x, o.y, o.z, o.w}; for (size_t k = 0; w * 16 + k < nc; k++) dst[k] = (int8_t)(ov[k >> 2] >> (8 * (k & 3))); }
        return;
    }
    const size_t b = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (b * 4 >= nc) return;
    const unsigned v = in[(r0 + r) * ldb + b0 + b];
    int8_t *o = out + r * ld_out + b * 4;
    const unsigned lut = 0xFF020100u;                                  // code -> int8 {0, 1, 2, -1}
#pragma unroll
    for (int k = 0; k < 4; k++) if (b * 4 + k < nc) o[k] = (int8_t)(lut >> (8 * ((v >> (2 * k)) & 3)));
}
int launch_geno_unpack(sfg_ctx *ctx, const sfg_geno *g, size_t r0, size_t c0, size_t nr, size_t nc, int8_t *out, size_t ld_out) {
    if (!g->packed || (c0 & 3)) SFG_FAIL(ctx, "geno_unpack: not a packed matrix or unaligned column");
    if (!nr || !nc) return 0;
    const bool vec = ((reinterpret_cast<uintptr_t>(g->dev) | g->ld | (c0 / 4)) & 3) == 0 && ((reinterpret_cast<uintptr_t>(out) | ld_out) & 15) == 0;
    if (vec) hipLaunchKernelGGL(k_geno_unpack<true>, dim3((unsigned)((nc + 4095) / 4096), (unsigned)nr), dim3(256), 0, ctx->stream, (const uint8_t *)g->dev, g->ld, r0, c0 / 4, nc, out, ld_out);
    else hipLaunchKernelGGL(k_geno_unpack<false>, dim3((unsigned)((nc + 1023) / 1024), (unsigned)nr), dim3(256), 0, ctx->stream, (const uint8_t *)g->dev, g->ld, r0, c0 / 4, nc, out, ld_out);
    SFG_HIP(ctx, hipGetLastError());
    return 0;
}
extern "C" int sfg_geno_pack(sfg_ctx *ctx, const sfg_geno *g, sfg_geno **out) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (g->packed) SFG_FAIL(ctx, "sfg_geno_pack: already packed");
    if (g->nrow > 65535u * 1024u) SFG_FAIL(ctx, "sfg_geno_pack: too many rows");
    const size_t ldw = (g->ncol + 15) / 16;
    unsigned *d = nullptr, *bad = nullptr, hbad = 0;
    SFG_HIP(ctx, hipMalloc(&d, g->nrow * ldw * 4));
    SFG_HIP(ctx, hipMalloc(&bad, 4));
    SFG_HIP(ctx, hipMemsetAsync(bad, 0, 4, ctx->stream));
    for (size_t r0 = 0; r0 < g->nrow; r0 += 65535) {
        const size_t nr = std::min<size_t>(65535, g->nrow - r0);
        hipLaunchKernelGGL(k_geno_pack, dim3((unsigned)((ldw + 255) / 256), (unsigned)nr), dim3(256), 0, ctx->stream, g->dev + r0 * g->ld, g->ncol, g->ld, d + r0 * ldw, ldw, bad);
    }
    SFG_HIP(ctx, hipGetLastError());
    SFG_HIP(ctx, hipMemcpyAsync(&hbad, bad, 4, hipMemcpyDeviceToHost, ctx->stream));
    SFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(bad);
    if (hbad) { (void)hipFree(d); SFG_FAIL(ctx, "sfg_geno_pack: %u values above 2 do not fit the 2-bit layout", hbad); }
    sfg_geno *p = new sfg_geno(); p->dev = (const int8_t *)d; p->nrow = g->nrow; p->ncol = g->ncol; p->ld = ldw * 4; p->owned = true; p->packed = true;
    *out = p; return 0;
}
extern "C" int sfg_geno_unpack(sfg_ctx *ctx, const sfg_geno *g, sfg_geno **out) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (!g->packed) SFG_FAIL(ctx, "sfg_geno_unpack: not packed");
    int8_t *d = nullptr;
    SFG_HIP(ctx, hipMalloc(&d, g->nrow * g->ncol));
    for (size_t r0 = 0; r0 < g->nrow; r0 += 65535) SFG_TRY(launch_geno_unpack(ctx, g, r0, 0, std::min<size_t>(65535, g->nrow - r0), g->ncol, d + r0 * g->ncol, g->ncol));
    SFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    sfg_geno *u = new sfg_geno(); u->dev = d; u->nrow = g->nrow; u->ncol = g->ncol; u->ld = g->ncol; u->owned = true;
    *out = u; return 0;
}
static int make_map(sfg_ctx *ctx, const uint8_t *filt, size_t n, int32_t **dev, size_t *kept) {
    *dev = nullptr; *kept = n;
    if (!filt) return 0;
    std::vector<int32_t> m(n); size_t k = 0;
    for (size_t i = 0; i < n; i++) m[i] = filt[i] ? (int32_t)k++ : -1;
    *kept = k;
    SFG_HIP(ctx, hipMalloc(dev, n * sizeof(int32_t)));
    SFG_HIP(ctx, hipMemcpy(*dev, m.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
    return 0;
}

extern "C" int sfg_geno_from_bed(sfg_ctx *ctx, const uint8_t *bed_host, size_t bed_bytes, size_t num_sample, size_t num_snp,
                                 const uint8_t *row_filter, const uint8_t *col_filter, sfg_geno **out) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bps = (num_sample + 3) / 4;
    if (!num_sample || !num_snp) SFG_FAIL(ctx, "sfg_geno_from_bed: bad dimensions");
    if (bed_bytes != 3 + num_snp * bps) SFG_FAIL(ctx, "sfg_geno_from_bed: file holds %zu bytes, expected 3 + %zu x %zu", bed_bytes, num_snp, bps);  // the script's assert
    if (bed_host[0] != 0x6C || bed_host[1] != 0x1B || bed_host[2] != 0x01) SFG_FAIL(ctx, "sfg_geno_from_bed: not a SNP-major PLINK .bed (magic %02x %02x %02x)", bed_host[0], bed_host[1], bed_host[2]);
    if (num_sample >= (1ULL << 31) || num_snp >= (1ULL << 31)) SFG_FAIL(ctx, "sfg_geno_from_bed: dimension too large");
    int32_t *rmap = nullptr, *cmap = nullptr; size_t nr, nc;
    SFG_TRY(make_map(ctx, row_filter, num_sample, &rmap, &nr));
    SFG_TRY(make_map(ctx, col_filter, num_snp, &cmap, &nc));
    if (!nr || !nc) { (void)hipFree(rmap); (void)hipFree(cmap); SFG_FAIL(ctx, "sfg_geno_from_bed: filters keep nothing"); }
    uint8_t *dbed = nullptr; int8_t *d = nullptr;
    SFG_HIP(ctx, hipMalloc(&dbed, num_snp * bps));
    SFG_HIP(ctx, hipMalloc(&d, nr * nc));
    SFG_HIP(ctx, hipMemcpyAsync(dbed, bed_host + 3, num_snp * bps, hipMemcpyHostToDevice, ctx->stream));
    hipLaunchKernelGGL(k_bed_decode, dim3((unsigned)((bps + 63) / 64), (unsigned)((num_snp + 63) / 64)), dim3(256), 0, ctx->stream,
                       dbed, bps, num_sample, num_snp, rmap, cmap, d, nc, BED_LUT);
    SFG_HIP(ctx, hipGetLastError());
    SFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(dbed); (void)hipFree(rmap); (void)hipFree(cmap);
    sfg_geno *g = new sfg_geno(); g->dev = d; g->nrow = nr; g->ncol = nc; g->ld = nc; g->owned = true;
    *out = g; return 0;
}

extern "C" int sfg_geno_dims(const sfg_geno *g, size_t *nrow, size_t *ncol) {
    if (!g) return 1;
    if (nrow) *nrow = g->nrow;
    if (ncol) *ncol = g->ncol;
    return 0;
}
extern "C" int sfg_geno_download(sfg_ctx *ctx, const sfg_geno *g, int8_t *host) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (g->packed) { sfg_geno *u = nullptr; SFG_TRY(sfg_geno_unpack(ctx, g, &u)); int rc = sfg_geno_download(ctx, u, host); sfg_geno_free(ctx, u); return rc; }
    SFG_HIP(ctx, hipMemcpy2DAsync(host, g->ncol, g->dev, g->ld, g->ncol, g->nrow, hipMemcpyDeviceToHost, ctx->stream));
    SFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
extern "C" int sfg_geno_transpose(sfg_ctx *ctx, const sfg_geno *g, sfg_geno **out) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (g->packed) SFG_FAIL(ctx, "sfg_geno_transpose: packed matrix (the products take the transpose flag on the one copy; sfg_geno_unpack first for a materialised transpose)");
    int8_t *d = nullptr;
    SFG_HIP(ctx, hipMalloc(&d, g->nrow * g->ncol));
    hipLaunchKernelGGL(k_geno_transpose, dim3((unsigned)((g->ncol + 63) / 64), (unsigned)((g->nrow + 63) / 64)), dim3(256), 0, ctx->stream, g->dev, g->nrow, g->ncol, g->ld, d);
    SFG_HIP(ctx, hipGetLastError());
    SFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    sfg_geno *t = new sfg_geno(); t->dev = d; t->nrow = g->ncol; t->ncol = g->nrow; t->ld = g->nrow; t->owned = true;
    *out = t; return 0;
}
extern "C" int sfg_geno_concat_cols(sfg_ctx *ctx, const sfg_geno *const *parts, int k, sfg_geno **out) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (k < 1) SFG_FAIL(ctx, "sfg_geno_concat_cols: nothing to merge");
    size_t nrow = parts[0]->nrow, ncol = 0;
    for (int i = 0; i < k; i++) if (parts[i]->packed) SFG_FAIL(ctx, "sfg_geno_concat_cols: part %d is packed (merge before sfg_geno_pack)", i);
    for (int i = 0; i < k; i++) { if (parts[i]->nrow != nrow) SFG_FAIL(ctx, "sfg_geno_concat_cols: part %d has %zu rows, expected %zu", i, parts[i]->nrow, nrow); ncol += parts[i]->ncol; }
    int8_t *d = nullptr;
    SFG_HIP(ctx, hipMalloc(&d, nrow * ncol));
    size_t c0 = 0;
    for (int i = 0; i < k; i++) {
        SFG_HIP(ctx, hipMemcpy2DAsync(d + c0, ncol, parts[i]->dev, parts[i]->ld, parts[i]->ncol, nrow, hipMemcpyDeviceToDevice, ctx->stream));
        c0 += parts[i]->ncol;
    }
    SFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    sfg_geno *g = new sfg_geno(); g->dev = d; g->nrow = nrow; g->ncol = ncol; g->ld = ncol; g->owned = true;
    *out = g; return 0;
}
