// pgen.hip — PLINK 2 .pgen hard calls decoded on the device (BASELINE config 1 runs on example_data/party*/geno/chrN.pgen).
// The reference never parses .pgen itself: gwas/utilities.go:141 FilterMatrixFilePgen shells out, per 8192-SNP batch (assoc.go:389), to
// scripts/filterMatrixPgen.sh:12-18 (plink2 --pfile --keep --extract --make-bed, then plinkBedToBinary.py), and its QC reads genotype counts that
// scripts/preprocessing/computeGenoCounts.py made with plink2 --geno-counts (qualcontrol.go:595).  plink2 exists neither here nor on a GPU box, so this
// file restates the PUBLISHED PGEN specification (plink-ng 2.0 pgenlib / pgen_spec): storage mode 0x10 (variable-width records; main-track types
// 0 = 2-bit, 1 = 1-bit + difflist, 2 / 3 = LD-compressed (3: 0 <-> 2 inverted afterwards), 4 / 6 / 7 = difflist over all-0 / all-2 / all-missing) and 0x02
// (fixed-width 2-bit).  A genotype code is the ALT allele count, 3 = missing - exactly the int8 value the reference's converters produce from plink2's
// .bed (BED 00 = hom A1 = ALT -> 2, 10 -> 1, 11 -> 0, 01 -> -1).
//
// The file bytes of a variant window cross PCIe as they are (<= 0.25 B per genotype); one workgroup decodes one variant record into a variant-major 2-bit
// row (difflist groups of 64 entries are independent: one thread each, patches by 32-bit atomics on disjoint bit pairs), LD-compressed records in a second
// pass from their base row; the rows then go through the .bed pipeline's transposing kernel (k_bed_decode with the PGEN code table) with the sample /
// variant filters fused, or are counted per variant (plink2 --geno-counts).  Parity: record types 0 and 1 are pinned by the reference's own fixture
// all.gcount.transpose.bin (tests/test_pgen.py); the other types are unpinned (no reference data uses them).
#include "common.hpp"
#include "kernels.hpp"
#include <algorithm>

#include "pgen.hpp"

// bytes of the header (magic .. end of the per-block tables) from the first 12 bytes of a file; 0 = not a supported .pgen
size_t pgen_header_bytes(const uint8_t *f12) {
    if (f12[0] != 0x6C || f12[1] != 0x1B) return 0;
    if (f12[2] == 0x02) return 12;
    if (f12[2] != 0x10) return 0;
    const uint64_t nv = (uint64_t)f12[3] | (uint64_t)f12[4] << 8 | (uint64_t)f12[5] << 16 | (uint64_t)f12[6] << 24;
    const unsigned ctrl = f12[11], wmode = ctrl & 15, ac_bytes = (ctrl >> 4) & 3, nonref = ctrl >> 6;
    if (wmode > 7) return 0;
    const unsigned vbits = wmode < 4 ? 4 : 8, lb = (wmode & 3) + 1;
    const uint64_t nblk = (nv + 65535) / 65536;
    size_t p = 12 + 8 * nblk;
    for (uint64_t b = 0; b < nblk; b++) {
        const uint64_t cnt = std::min<uint64_t>(65536, nv - b * 65536);
        p += (vbits == 4 ? (cnt + 1) / 2 : cnt) + cnt * lb + cnt * ac_bytes + (nonref == 3 ? (cnt + 7) / 8 : 0);
    }
    return p;
}
// f: at least the header bytes (`bytes` of them are readable); file_bytes: the size of the whole file, against which the record table is validated
static int pgen_index_unchecked(sfg_ctx *ctx, const uint8_t *f, size_t bytes, size_t file_bytes, PgenIndex &ix) {
    if (!f || bytes < 12 || f[0] != 0x6C || f[1] != 0x1B) SFG_FAIL(ctx, "pgen: not a PLINK 2 .pgen (bad magic)");
    if (bytes > file_bytes) bytes = file_bytes;
    auto u32 = [&](size_t p) { return (uint32_t)f[p] | (uint32_t)f[p + 1] << 8 | (uint32_t)f[p + 2] << 16 | (uint32_t)f[p + 3] << 24; };
    ix.nv = u32(3); ix.ns = u32(7);
    if (!ix.nv || !ix.ns) SFG_FAIL(ctx, "pgen: empty file (%u variants, %u samples)", ix.nv, ix.ns);
    if (ix.ns > 0x7FFFFFFFu) SFG_FAIL(ctx, "pgen: sample count %u out of range", ix.ns);
    // a hostile / corrupt header must not size anything: the variant count is believed only as far as the file is long enough to hold that many entries
    if (f[2] == 0x02) {
        const uint64_t bps = ((uint64_t)ix.ns + 3) / 4;
        if ((file_bytes - 12) / bps < ix.nv) SFG_FAIL(ctx, "pgen: truncated fixed-width file");
        ix.off.resize(ix.nv); ix.len.resize(ix.nv); ix.vrt.resize(ix.nv);
        for (uint32_t v = 0; v < ix.nv; v++) { ix.off[v] = 12 + (uint64_t)v * bps; ix.len[v] = (uint32_t)bps; ix.vrt[v] = 0; }
        return 0;
    }
    if (f[2] != 0x10) SFG_FAIL(ctx, "pgen: storage mode 0x%02x not supported (0x10 variable-width and 0x02 fixed-width 2-bit are)", f[2]);
    const unsigned ctrl = f[11], wmode = ctrl & 15, ac_bytes = (ctrl >> 4) & 3, nonref = ctrl >> 6;
    if (wmode > 7) SFG_FAIL(ctx, "pgen: header control byte 0x%02x not supported", ctrl);
    const unsigned vbits = wmode < 4 ? 4 : 8, lb = (wmode & 3) + 1;
    const uint32_t nblk = (ix.nv + 65535) / 65536;
    const size_t header_end = pgen_header_bytes(f);                       // 12 + block offsets + per-block tables: >= 1.5 bytes per variant
    if (!header_end || header_end > file_bytes) SFG_FAIL(ctx, "pgen: the header tables of %u variants do not fit the file (%zu bytes)", ix.nv, file_bytes);
    if (header_end > bytes) SFG_FAIL(ctx, "pgen: truncated header");
    ix.off.resize(ix.nv); ix.len.resize(ix.nv); ix.vrt.resize(ix.nv);
    size_t p = 12 + (size_t)8 * nblk;
    uint64_t cur = header_end;                                            // records follow the header, block after block, in file order
    for (uint32_t b = 0; b < nblk; b++) {
        const uint32_t v0 = b * 65536u, cnt = std::min<uint32_t>(65536u, ix.nv - v0);
        uint64_t bo = 0; for (int k = 0; k < 8; k++) bo |= (uint64_t)f[12 + 8 * (size_t)b + k] << (8 * k);
        if (bo < header_end || bo < cur || bo > file_bytes) SFG_FAIL(ctx, "pgen: block %u starts at byte %llu, outside [%llu, %zu] (header end / previous block end, file size)", b, (unsigned long long)bo, (unsigned long long)cur, file_bytes);
        cur = bo;
        const size_t vt_bytes = vbits == 4 ? (cnt + 1) / 2 : cnt;
        if (p + vt_bytes + (size_t)cnt * lb + (size_t)cnt * ac_bytes > bytes) SFG_FAIL(ctx, "pgen: truncated header");
        for (uint32_t k = 0; k < cnt; k++) ix.vrt[v0 + k] = vbits == 4 ? (uint8_t)((f[p + k / 2] >> (4 * (k & 1))) & 15) : f[p + k];
        p += vt_bytes;
        for (uint32_t k = 0; k < cnt; k++) {
            uint32_t x = 0; for (unsigned j = 0; j < lb; j++) x |= (uint32_t)f[p + (size_t)k * lb + j] << (8 * j);
            ix.len[v0 + k] = x; ix.off[v0 + k] = cur;
            if (x > file_bytes - cur) SFG_FAIL(ctx, "pgen: record %u runs past the end of the file", v0 + k);      // cur <= file_bytes: no wrap
            cur += x;
        }
        p += (size_t)cnt * lb + (size_t)cnt * ac_bytes;
        if (nonref == 3) p += (cnt + 7) / 8;
    }
    return 0;
}
// f: at least the header bytes (`bytes` of them are readable); file_bytes: the size of the whole file, against which the record table is validated.
// Nothing thrown by the containers (std::bad_alloc, std::length_error) crosses the extern "C" boundary of the callers.
int pgen_index(sfg_ctx *ctx, const uint8_t *f, size_t bytes, size_t file_bytes, PgenIndex &ix) {
    try { return pgen_index_unchecked(ctx, f, bytes, file_bytes, ix); }
    catch (const std::exception &e) { SFG_FAIL(ctx, "pgen: cannot index the file (%s)", e.what()); }
}

enum { PGEN_ERR_FORMAT = 1, PGEN_ERR_MULTIALLELIC = 2, PGEN_ERR_TYPE = 4 };

__device__ __forceinline__ bool pg_varint(const uint8_t *&p, const uint8_t *end, uint32_t &out) {
    uint32_t v = 0; int sh = 0;
    while (p < end && sh < 35) { const uint32_t b = *p++; v |= (b & 0x7Fu) << sh; if (!(b & 0x80u)) { out = v; return true; } sh += 7; }
    return false;
}
__device__ __forceinline__ void pg_patch(unsigned *row, uint32_t id, unsigned g) {
    const int sh = 2 * (int)(id & 15);
    atomicAnd(&row[id >> 4], ~(3u << sh));
    atomicOr(&row[id >> 4], g << sh);
}

// one workgroup per variant row of the window; pass 0: records that stand alone, pass 1: LD-compressed records (their base row is complete by then).
// file: the window's bytes (record r at off[r], len[r] bytes).  rows: [nrows][pitch] bytes, pitch a multiple of 4.
__global__ void __launch_bounds__(256) k_pgen_decode(const uint8_t *file, const uint64_t *off, const uint32_t *len, const uint8_t *vrt, const uint32_t *ldbase,
                                                     uint32_t ns, size_t pitch, uint8_t *rows, int pass, int *err) {
    const uint32_t r = blockIdx.x; const int tid = threadIdx.x;
    const unsigned vt = vrt[r], mt = vt & 7;
    if (((mt == 2 || mt == 3) ? 1 : 0) != pass) return;
    if (vt & 8) { if (!tid) atomicOr(err, PGEN_ERR_MULTIALLELIC); return; }        // multiallelic hard calls: plink2 --make-bed refuses those too
    if (mt == 5) { if (!tid) atomicOr(err, PGEN_ERR_TYPE); return; }
    unsigned *row = reinterpret_cast<unsigned *>(rows + (size_t)r * pitch);
    uint8_t *rowb = rows + (size_t)r * pitch;
    const size_t bps = ((size_t)ns + 3) / 4, words = pitch / 4;
    const uint8_t *p = file + off[r], *end = p + len[r];
    __shared__ uint32_t sh_len; __shared__ const uint8_t *sh_first, *sh_rare, *sh_delta;
    if (mt == 0) {
        if (len[r] < bps) { if (!tid) atomicOr(err, PGEN_ERR_FORMAT); return; }
        for (size_t b = tid; b < pitch; b += 256) rowb[b] = b < bps ? p[b] : (uint8_t)0;
    } else if (mt == 1) {
        const size_t bitbytes = ((size_t)ns + 7) / 8;
        if ((size_t)len[r] < 1 + bitbytes) { if (!tid) atomicOr(err, PGEN_ERR_FORMAT); return; }
        const unsigned code = p[0], lo = code >> 2, delta = code & 3;
        if (!delta || lo + delta > 3) { if (!tid) atomicOr(err, PGEN_ERR_FORMAT); return; }
        const uint8_t *bits = p + 1;
        for (size_t b = tid; b < pitch; b += 256) {
            unsigned o = 0;
            if (b < bps) {
                const unsigned nib = (bits[b >> 1] >> (4 * (b & 1))) & 15u;
#pragma unroll
                for (int k = 0; k < 4; k++) o |= (lo + delta * ((nib >> k) & 1u)) << (2 * k);
            }
            rowb[b] = (uint8_t)o;
        }
        p = bits + bitbytes;
    } else if (mt == 2 || mt == 3) {
        const unsigned *base = reinterpret_cast<const unsigned *>(rows + (size_t)ldbase[r] * pitch);
        for (size_t w = tid; w < words; w += 256) row[w] = base[w];
    } else {
        const unsigned fill = (mt & 3u) * 0x55555555u;                               // 4 -> 0, 6 -> 2, 7 -> 3 (missing)
        for (size_t w = tid; w < words; w += 256) row[w] = fill;
    }
    if (mt != 0) {
        if (!tid) {
            uint32_t n = 0; const uint8_t *q = p; bool ok = pg_varint(q, end, n) && n <= ns;
            if (ok && n) {
                int idb = 1; while (idb < 4 && (ns >> (8 * idb))) idb++;
                const uint32_t gc = (n + 63) / 64;
                sh_first = q; sh_rare = q + (size_t)gc * idb + (gc - 1); sh_delta = sh_rare + (n + 3) / 4;
                ok = sh_delta <= end;
            }
            sh_len = ok ? n : 0;
            if (!ok) atomicOr(err, PGEN_ERR_FORMAT);
        }
        __syncthreads();                                                             // also orders the fill above before the patches
        const uint32_t n = sh_len;
        if (n) {
            int idb = 1; while (idb < 4 && (ns >> (8 * idb))) idb++;
            const uint32_t gc = (n + 63) / 64;
            const uint8_t *sizes = sh_first + (size_t)gc * idb;
            for (uint32_t g = tid; g < gc; g += 256) {                               // a group of 64 entries is self-contained
                size_t doff = 0; for (uint32_t h = 0; h < g; h++) doff += 63u + sizes[h];
                const uint8_t *q = sh_delta + doff;
                uint32_t id = 0; for (int b = 0; b < idb; b++) id |= (uint32_t)sh_first[(size_t)g * idb + b] << (8 * b);
                const uint32_t k0 = g * 64, k1 = k0 + 64 < n ? k0 + 64 : n;
                bool ok = true;
                for (uint32_t k = k0; k < k1 && ok; k++) {
                    if (k > k0) { uint32_t dl; ok = pg_varint(q, end, dl); id += dl; }
                    ok = ok && id < ns;
                    if (ok) pg_patch(row, id, (sh_rare[k >> 2] >> (2 * (k & 3))) & 3u);
                }
                if (!ok) atomicOr(err, PGEN_ERR_FORMAT);
            }
        }
        __syncthreads();
    } else __syncthreads();
    // type 3: 0 <-> 2 after the difflist; then clear the padding codes of the last dwords
    for (size_t w = tid; w < words; w += 256) {
        unsigned x = row[w];
        if (mt == 3) x ^= ((~x) << 1) & 0xAAAAAAAAu;
        const size_t s0 = w * 16;
        if (s0 + 16 > ns) x = s0 >= ns ? 0u : (x & ((1u << (2 * (ns - s0))) - 1u));
        row[w] = x;
    }
}

// plink2 --geno-counts for diploid hard calls: counts[0..5][nv] = HOM_REF_CT, HET_REF_ALT_CTS, TWO_ALT_GENO_CTS, HAP_REF_CT, HAP_ALT_CTS, MISSING_CT.
// keep: [pitch/4] dwords with 01 in the bit pair of every counted sample
__global__ void __launch_bounds__(256) k_pgen_counts(const uint8_t *rows, size_t pitch, const unsigned *keep, uint32_t nv, uint32_t *counts) {
    const uint32_t v = blockIdx.x; const int tid = threadIdx.x;
    const unsigned *row = reinterpret_cast<const unsigned *>(rows + (size_t)v * pitch);
    unsigned c1 = 0, c2 = 0, c3 = 0, ct = 0;
    for (size_t w = tid; w < pitch / 4; w += 256) {
        const unsigned x = row[w], m = keep[w], lo = x & m, hi = (x >> 1) & m;
        c1 += __popc(lo & ~hi); c2 += __popc(hi & ~lo); c3 += __popc(lo & hi); ct += __popc(m);
    }
    __shared__ unsigned red[4];
    if (tid < 4) red[tid] = 0;
    __syncthreads();
    atomicAdd(&red[0], ct - c1 - c2 - c3); atomicAdd(&red[1], c1); atomicAdd(&red[2], c2); atomicAdd(&red[3], c3);
    __syncthreads();
    if (!tid) {
        counts[v] = red[0]; counts[(size_t)nv + v] = red[1]; counts[(size_t)2 * nv + v] = red[2];
        counts[(size_t)3 * nv + v] = 0; counts[(size_t)4 * nv + v] = 0; counts[(size_t)5 * nv + v] = red[3];
    }
}

int launch_bed_decode_lut(sfg_ctx *ctx, hipStream_t st, const uint8_t *dbed, size_t bps, size_t num_sample, size_t num_snp, const int32_t *rmap, const int32_t *cmap,
                          int8_t *out, size_t ld, unsigned lut);

// the variants a window [v0, v1) needs on the device: [start, v1) with start <= v0 the LD base of the first ones; record offsets relative to the window's
// first byte f0, LD-base row per record
int pgen_window(sfg_ctx *ctx, const PgenIndex &ix, size_t file_bytes, size_t v0, size_t v1, PgenWindow &w) {
    if (v0 >= v1 || v1 > ix.nv) SFG_FAIL(ctx, "pgen: variant range [%zu, %zu) out of bounds (%u variants)", v0, v1, ix.nv);
    size_t start = v0;
    while (start > 0 && (ix.vrt[start] & 6) == 2) start--;
    if ((ix.vrt[start] & 6) == 2) SFG_FAIL(ctx, "pgen: the first variant is LD-compressed");
    w.start = start; w.nr = v1 - start; w.lead = v0 - start;
    w.f0 = ix.off[start]; w.f1 = ix.off[v1 - 1] + ix.len[v1 - 1];
    if (w.f1 > file_bytes || w.f0 > w.f1) SFG_FAIL(ctx, "pgen: record table inconsistent with the file size");
    w.off.resize(w.nr); w.ldb.assign(w.nr, 0xFFFFFFFFu); uint32_t last = 0;
    for (size_t r = 0; r < w.nr; r++) {
        const uint64_t o = ix.off[start + r], l = ix.len[start + r];
        if (o < w.f0 || o > w.f1 || l > w.f1 - o) SFG_FAIL(ctx, "pgen: record %zu lies outside its window's byte range", start + r);     // every record, not only the first and last
        w.off[r] = ix.off[start + r] - w.f0;
        if ((ix.vrt[start + r] & 6) == 2) w.ldb[r] = last; else last = (uint32_t)r;
    }
    return 0;
}
size_t pgen_pitch(const PgenIndex &ix) { return ((((size_t)ix.ns + 3) / 4) + 3) & ~(size_t)3; }
// descriptor block of a window on the device: off | len | vrt | ldbase | err, `bytes` = pgen_desc_bytes(nr)
size_t pgen_desc_bytes(size_t nr) { auto al = [](size_t x) { return (x + 255) & ~(size_t)255; }; return al(nr * 8) + al(nr * 4) + al(nr) + al(nr * 4) + 256; }
int pgen_upload_desc(sfg_ctx *ctx, hipStream_t st, const PgenIndex &ix, const PgenWindow &w, uint8_t *desc) {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t nr = w.nr, o_len = al(nr * 8), o_vrt = o_len + al(nr * 4), o_ldb = o_vrt + al(nr), o_err = o_ldb + al(nr * 4);
    SFG_HIP(ctx, hipMemcpyAsync(desc, w.off.data(), nr * 8, hipMemcpyHostToDevice, st));
    SFG_HIP(ctx, hipMemcpyAsync(desc + o_len, ix.len.data() + w.start, nr * 4, hipMemcpyHostToDevice, st));
    SFG_HIP(ctx, hipMemcpyAsync(desc + o_vrt, ix.vrt.data() + w.start, nr, hipMemcpyHostToDevice, st));
    SFG_HIP(ctx, hipMemcpyAsync(desc + o_ldb, w.ldb.data(), nr * 4, hipMemcpyHostToDevice, st));
    SFG_HIP(ctx, hipMemsetAsync(desc + o_err, 0, 4, st));
    return 0;
}
// the two decode passes of a window whose bytes and descriptors are on the device; rows: [nr][pitch].  *err_dev (returned) holds the error flags afterwards
int launch_pgen_decode(sfg_ctx *ctx, hipStream_t st, const uint8_t *file_dev, const uint8_t *desc, size_t nr, uint32_t ns, size_t pitch, uint8_t *rows, const int **err_dev) {
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_len = al(nr * 8), o_vrt = o_len + al(nr * 4), o_ldb = o_vrt + al(nr), o_err = o_ldb + al(nr * 4);
    for (int pass = 0; pass < 2; pass++) {
        hipLaunchKernelGGL(k_pgen_decode, dim3((unsigned)nr), dim3(256), 0, st, file_dev, (const uint64_t *)desc, (const uint32_t *)(desc + o_len), desc + o_vrt,
                           (const uint32_t *)(desc + o_ldb), ns, pitch, rows, pass, (int *)(desc + o_err));
        SFG_HIP(ctx, hipGetLastError());
    }
    if (err_dev) *err_dev = (const int *)(desc + o_err);
    return 0;
}
int pgen_decode_error(sfg_ctx *ctx, int herr) {
    if (!herr) return 0;
    if (herr & PGEN_ERR_MULTIALLELIC) SFG_FAIL(ctx, "pgen: multiallelic hard calls present (plink2 --make-bed refuses them as well; split them first)");
    if (herr & PGEN_ERR_TYPE) SFG_FAIL(ctx, "pgen: record type 5 is not defined by the PGEN specification");
    SFG_FAIL(ctx, "pgen: malformed variant record");
}

// decodes variants [v0, v1) of a file image in host memory into device rows; returns the rows of [v0, v1)
struct PgenRows { uint8_t *buf = nullptr; uint8_t *rows = nullptr; size_t pitch = 0; };
static int pgen_decode_window(sfg_ctx *ctx, const uint8_t *f, size_t bytes, const PgenIndex &ix, size_t v0, size_t v1, PgenRows &out) {
    PgenWindow w; SFG_TRY(pgen_window(ctx, ix, bytes, v0, v1, w));
    const size_t pitch = pgen_pitch(ix), fb = (size_t)(w.f1 - w.f0);
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    const size_t o_file = al(w.nr * pitch), o_desc = o_file + al(fb + 8);            // one allocation: rows | file bytes | descriptors
    uint8_t *d = nullptr;
    SFG_HIP(ctx, hipMalloc(&d, o_desc + pgen_desc_bytes(w.nr)));
    int rc = 0, herr = 0; const int *err_dev = nullptr;
    if (hipMemcpyAsync(d + o_file, f + w.f0, fb, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = 1;
    if (!rc) rc = pgen_upload_desc(ctx, ctx->stream, ix, w, d + o_desc);
    if (!rc) rc = launch_pgen_decode(ctx, ctx->stream, d + o_file, d + o_desc, w.nr, ix.ns, pitch, d, &err_dev);
    if (!rc && (hipMemcpyAsync(&herr, err_dev, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)) rc = 1;   // the host vectors are done with, too
    if (rc) { (void)hipFree(d); if (ctx->err.empty()) ctx->err = "pgen: device decode failed to launch"; return 1; }
    if (pgen_decode_error(ctx, herr)) { (void)hipFree(d); return 1; }
    out.buf = d; out.rows = d + w.lead * pitch; out.pitch = pitch;
    return 0;
}

extern "C" int sfg_pgen_dims(sfg_ctx *ctx, const uint8_t *pgen_host, size_t pgen_bytes, size_t *num_sample, size_t *num_variant) {
    PgenIndex ix; SFG_TRY(pgen_index(ctx, pgen_host, pgen_bytes, pgen_bytes, ix));
    if (num_sample) *num_sample = ix.ns;
    if (num_variant) *num_variant = ix.nv;
    return 0;
}

static int make_map32(sfg_ctx *ctx, const uint8_t *filt, size_t n, int32_t **dev, size_t *kept) {
    *dev = nullptr; *kept = n;
    if (!filt) return 0;
    std::vector<int32_t> m(n); size_t k = 0;
    for (size_t i = 0; i < n; i++) m[i] = filt[i] ? (int32_t)k++ : -1;
    *kept = k;
    SFG_HIP(ctx, hipMalloc(dev, n * sizeof(int32_t)));
    SFG_HIP(ctx, hipMemcpy(*dev, m.data(), n * sizeof(int32_t), hipMemcpyHostToDevice));
    return 0;
}

extern "C" int sfg_geno_from_pgen(sfg_ctx *ctx, const uint8_t *pgen_host, size_t pgen_bytes, size_t v0, size_t v1,
                                  const uint8_t *row_filter, const uint8_t *col_filter, sfg_geno **out) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    PgenIndex ix; SFG_TRY(pgen_index(ctx, pgen_host, pgen_bytes, pgen_bytes, ix));
    if (v1 == 0) v1 = ix.nv;                                       // [0, 0) = the whole file
    PgenRows rows; SFG_TRY(pgen_decode_window(ctx, pgen_host, pgen_bytes, ix, v0, v1, rows));
    int32_t *rmap = nullptr, *cmap = nullptr; size_t nr = 0, nc = 0; int8_t *d = nullptr;
    int rc = make_map32(ctx, row_filter, ix.ns, &rmap, &nr);
    if (!rc) rc = make_map32(ctx, col_filter, v1 - v0, &cmap, &nc);
    if (!rc && (!nr || !nc)) { ctx->err = "sfg_geno_from_pgen: filters keep nothing"; rc = 1; }
    if (!rc && hipMalloc(&d, nr * nc) != hipSuccess) { ctx->err = "sfg_geno_from_pgen: out of device memory"; rc = 1; }
    if (!rc) rc = launch_bed_decode_lut(ctx, ctx->stream, rows.rows, rows.pitch, ix.ns, v1 - v0, rmap, cmap, d, nc, 0xFF020100u);     // code -> int8 {0, 1, 2, -1}
    if (!rc && hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->err = "sfg_geno_from_pgen: decode failed"; rc = 1; }
    (void)hipFree(rows.buf); (void)hipFree(rmap); (void)hipFree(cmap);
    if (rc) { (void)hipFree(d); return rc; }
    sfg_geno *g = new sfg_geno(); g->dev = d; g->nrow = nr; g->ncol = nc; g->ld = nc; g->owned = true;
    *out = g; return 0;
}

extern "C" int sfg_pgen_geno_counts(sfg_ctx *ctx, const uint8_t *pgen_host, size_t pgen_bytes, const uint8_t *row_filter, uint32_t *counts_host) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    PgenIndex ix; SFG_TRY(pgen_index(ctx, pgen_host, pgen_bytes, pgen_bytes, ix));
    PgenRows rows; SFG_TRY(pgen_decode_window(ctx, pgen_host, pgen_bytes, ix, 0, ix.nv, rows));
    std::vector<unsigned> keep(rows.pitch / 4, 0u);
    for (uint32_t i = 0; i < ix.ns; i++) if (!row_filter || row_filter[i]) keep[i >> 4] |= 1u << (2 * (i & 15));
    unsigned *dk = nullptr; uint32_t *dc = nullptr; int rc = 0;
    if (hipMalloc(&dk, keep.size() * 4) != hipSuccess || hipMalloc(&dc, (size_t)6 * ix.nv * 4) != hipSuccess) { ctx->err = "sfg_pgen_geno_counts: out of device memory"; rc = 1; }
    if (!rc && hipMemcpyAsync(dk, keep.data(), keep.size() * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = 1;
    if (!rc) { hipLaunchKernelGGL(k_pgen_counts, dim3(ix.nv), dim3(256), 0, ctx->stream, rows.rows, rows.pitch, dk, ix.nv, dc); if (hipGetLastError() != hipSuccess) rc = 1; }
    if (!rc && (hipMemcpyAsync(counts_host, dc, (size_t)6 * ix.nv * 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)) rc = 1;
    if (rc && ctx->err.empty()) ctx->err = "sfg_pgen_geno_counts: device work failed";
    (void)hipFree(rows.buf); (void)hipFree(dk); (void)hipFree(dc);
    return rc;
}

// GenoBlockMult over one chromosome's .pgen (gwas/assoc.go:340-420) - what config 1 runs: per batch of `batch_snps` KEPT variants FilterMatrixFilePgen +
// NewGenoFileStream + MatMult4Stream(cps, mat, X, 5, false, square, nproc), outputs concatenated (crypto.ConcatCipherMatrix).  pgen_host: the file image
// (mmap it for large files); row_filter: one byte per sample (the --keep list), col_filter: one byte per variant (snpFilt), NULL = keep all.
// Same output layout and column sums as sfg_assoc_stream_bed; the baby-step rotation cache of A is built once for all batches.
extern "C" int sfg_assoc_pgen(sfg_ctx *ctx, const uint8_t *pgen_host, size_t pgen_bytes, const uint8_t *row_filter, const uint8_t *col_filter, size_t batch_snps,
                              const uint64_t *A_dev, int s, int in_level, int max_level, unsigned flags,
                              uint64_t *out_dev, size_t out_ct_capacity, size_t *out_ct, double *sum_host, double *sqsum_host) {
    ApiScope api_scope(ctx);
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (!batch_snps) SFG_FAIL(ctx, "assoc_pgen: bad batch size");
    if (flags & SFG_TRANSPOSE) SFG_FAIL(ctx, "assoc_pgen: batches are multiplied as X (samples x SNPs)");
    PgenIndex ix; SFG_TRY(pgen_index(ctx, pgen_host, pgen_bytes, pgen_bytes, ix));
    const size_t slots = SFG_SLOTS, N = SFG_N, ctw = (size_t)2 * max_level * N;
    struct B { size_t v0, v1, kept; };
    std::vector<B> bt; size_t start = 0, counter = 0;                     // assoc.go:371-416: a batch closes at batch_snps kept variants or at the end of the file
    for (size_t idx = 0; idx < ix.nv; idx++) {
        if (!col_filter || col_filter[idx]) counter++;
        if (counter == batch_snps || (idx == (size_t)ix.nv - 1 && counter > 0)) { bt.push_back({start, idx + 1, counter}); start = idx + 1; counter = 0; }
    }
    size_t total_ct = 0, max_kept = 0, nr = 0;
    for (const B &b : bt) { total_ct += (b.kept + slots - 1) / slots; max_kept = std::max(max_kept, b.kept); }
    for (uint32_t i = 0; i < ix.ns; i++) nr += !row_filter || row_filter[i];
    if (out_ct) *out_ct = total_ct;
    if (bt.empty()) return 0;
    if (!nr) SFG_FAIL(ctx, "assoc_pgen: the row filter keeps nothing");
    if (total_ct > out_ct_capacity) SFG_FAIL(ctx, "assoc_pgen: output needs %zu ciphertexts per row, capacity %zu", total_ct, out_ct_capacity);
    std::vector<size_t> widths;
    for (const B &b : bt) for (size_t c0 = 0; c0 < b.kept; c0 += slots) { const size_t w = std::min(slots, b.kept - c0); if (std::find(widths.begin(), widths.end(), w) == widths.end()) widths.push_back(w); }
    AssocRot rot; u64 *tmp = nullptr;
    SFG_TRY(assoc_build_rot(ctx, (const u64 *)A_dev, s, in_level, max_level, nr, widths, rot));
    int rc = 0;
    if (hipMalloc(&tmp, (size_t)s * ((max_kept + slots - 1) / slots) * ctw * 8) != hipSuccess) { ctx->err = "assoc_pgen: out of device memory"; rc = 1; }
    size_t out_shift = 0;
    for (size_t k = 0; k < bt.size() && !rc; k++) {
        const B &b = bt[k];
        sfg_geno *g = nullptr;
        rc = sfg_geno_from_pgen(ctx, pgen_host, pgen_bytes, b.v0, b.v1, row_filter, col_filter ? col_filter + b.v0 : nullptr, &g);
        if (rc) break;
        const size_t nct = (b.kept + slots - 1) / slots;
        rc = assoc_product(ctx, rot, A_dev, s, in_level, max_level, g, flags, (int)nct, (uint64_t *)tmp);
        for (int i = 0; i < s && !rc; i++)
            if (hipMemcpyAsync(out_dev + ((size_t)i * out_ct_capacity + out_shift) * ctw, tmp + (size_t)i * nct * ctw, nct * ctw * 8, hipMemcpyDeviceToDevice, ctx->stream) != hipSuccess) { ctx->err = "assoc_pgen: copy failed"; rc = 1; }
        if (!rc && (sum_host || sqsum_host)) {
            if (sum_host) std::fill(sum_host + out_shift * slots, sum_host + (out_shift + nct) * slots, 0.0);
            if (sqsum_host) std::fill(sqsum_host + out_shift * slots, sqsum_host + (out_shift + nct) * slots, 0.0);
            rc = sfg_geno_colsums(ctx, g, sum_host ? sum_host + out_shift * slots : nullptr, sqsum_host ? sqsum_host + out_shift * slots : nullptr);
        }
        sfg_geno_free(ctx, g);                                            // synchronises the queue: tmp and the batch matrix are done with
        out_shift += nct;
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(tmp); assoc_free_rot(rot);
    return rc;
}
