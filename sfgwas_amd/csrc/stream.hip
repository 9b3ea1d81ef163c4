// stream.hip — the association scan's genotype batches streamed from storage (BASELINE config 5).
//
// Reference (gwas/assoc.go:340-420): GenoBlockMult walks the SNPs of a chromosome file; every `pgenBatchSize` KEPT SNPs it shells out to
// plink2 + Python (scripts/filterMatrixPgen.sh, plinkBedToBinary.py, transposeMatrix.py) to materialise an int8 [numInd x batch] temp file,
// opens it as a GenoFileStream, calls MatMult4Stream(cps, mat, X, 5, false, square, nproc) and concatenates the batch outputs
// (crypto.ConcatCipherMatrix).  Here the SNP-major 2-bit PLINK .bed IS the input: a batch is one contiguous byte range of the file
// (bps = ceil(num_sample / 4) bytes per SNP), read with pread() into pinned memory by a reader thread while the GPU works on the previous
// batch, copied to HBM as packed 2-bit codes (4x less than int8 over PCIe), decoded / filtered / transposed on the device
// (k_bed_decode, pinned by the reference scripts' outputs in tests/test_input_formats.py), multiplied, concatenated.  A .pgen file is
// converted once with `plink2 --make-bed` (the reference already depends on plink2 for this path), or decoded natively (pgen.hip).
//
// Every batch multiplies the SAME ciphertext matrix `mat` (GenoBlockMult(b, concat, ...) walks all chromosome batches with one concat, assoc.go:714-718;
// gWY's four calls per block likewise), and the reference recomputes rotCache[i][baby] = RotateRight(mat[i][bi], -baby) inside every MatMult4Stream call
// (matmult.go:1373-1377) - 90 key switches per input ciphertext, which at one block column per batch are the LARGEST item of a batch (47 % of the kernel
// time at 500 000 x 8192, s = 13).  The rotations depend on `mat` only, so one call builds the baby-step rotation cache once (the same key switches, the
// same bits; 1.86 GB per block row at s = 13: 115 GB for 500 000 samples) and every batch multiplies against it (SFG_ASSOC_ROTCACHE_MB=0 restores the
// per-batch rebuild; a cache that exceeds the budget falls back to it).
#include "common.hpp"
#include <chrono>
#include "kernels.hpp"
#include "pgen.hpp"
#include <algorithm>
#include <condition_variable>
#include <cerrno>
#include <fcntl.h>
#include <mutex>
#include <sys/stat.h>
#include <thread>
#include <unistd.h>

// genoio.hip
int launch_bed_decode(sfg_ctx *ctx, hipStream_t st, const uint8_t *dbed, size_t bps, size_t num_sample, size_t num_snp, const int32_t *rmap, const int32_t *cmap,
                      int8_t *out, size_t ld);
int launch_bed_decode_lut(sfg_ctx *ctx, hipStream_t st, const uint8_t *dbed, size_t bps, size_t num_sample, size_t num_snp, const int32_t *rmap, const int32_t *cmap,
                          int8_t *out, size_t ld, unsigned lut);

namespace {
struct Batch { size_t snp0, nsnp, kept; off_t off = 0; size_t bytes = 0; };            // file SNPs [snp0, snp0 + nsnp), `kept` of them pass the filter; their bytes in the file
// assoc.go:371-416: a batch closes when `batch_snps` kept SNPs have been seen or the file ends with a non-empty batch
std::vector<Batch> make_batches(const uint8_t *col_filter, size_t num_snp, size_t batch_snps) {
    std::vector<Batch> b; size_t start = 0, counter = 0;
    for (size_t idx = 0; idx < num_snp; idx++) {
        if (!col_filter || col_filter[idx]) counter++;
        if (counter == batch_snps || (idx == num_snp - 1 && counter > 0)) { b.push_back({start, idx + 1 - start, counter}); start = idx + 1; counter = 0; }
    }
    return b;
}
struct Reader {                                        // fills pinned slot k & 1 with the bytes of batch k, one batch ahead of the consumer
    int fd; const std::vector<Batch> *bt; uint8_t *slot[2];
    bool direct = false; size_t lead[2] = {0, 0};       // O_DIRECT: 4096-byte aligned file ranges; the batch starts `lead` bytes into its slot
    std::mutex mu; std::condition_variable cv; long filled = -1, released = -1; bool failed = false; std::string err;
    void run() {
        for (size_t k = 0; k < bt->size(); k++) {
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return (long)k - 2 <= released; }); }      // slot k & 1 was last used by batch k - 2
            const Batch &b = (*bt)[k]; const off_t off = b.off, a0 = direct ? off & ~(off_t)4095 : off;
            const size_t ld = (size_t)(off - a0), want = ld + b.bytes, want_al = direct ? (want + 4095) & ~(size_t)4095 : want; size_t got = 0;
            while (got < want) {
                ssize_t r = pread(fd, slot[k & 1] + got, want_al - got, a0 + (off_t)got);
                if (r <= 0) { std::lock_guard<std::mutex> lk(mu); failed = true; err = "short read from the genotype file"; cv.notify_all(); return; }
                got += (size_t)r;
            }
            lead[k & 1] = ld;
            { std::lock_guard<std::mutex> lk(mu); filled = (long)k; }
            cv.notify_all();
        }
    }
    bool wait_filled(size_t k) { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return failed || filled >= (long)k; }); return !failed; }
    void release(size_t k) { { std::lock_guard<std::mutex> lk(mu); released = (long)k; } cv.notify_all(); }
};
}  // namespace

static void assoc_baby_tabs(size_t nr, const std::vector<size_t> &widths, std::vector<std::vector<uint8_t>> &tabs) {
    const size_t slots = SFG_SLOTS;
    const int nbr = (int)((nr + slots - 1) / slots);
    tabs.assign(nbr, std::vector<uint8_t>(SFG_D, 0));
    for (int bi = 0; bi < nbr; bi++) {
        const int rows = (int)(std::min((size_t)(bi + 1) * slots, nr) - (size_t)bi * slots);
        for (int shift = 0; shift < SFG_SLOTS; shift++) {
            if (tabs[bi][shift % SFG_D]) continue;
            for (size_t w : widths) if (sfg_diag_bool(rows, (int)w, SFG_SLOTS, -shift)) { tabs[bi][shift % SFG_D] = 1; break; }
        }
    }
}
// The baby-step rotation cache of the ciphertext matrix every batch of an association scan multiplies (see the header comment): *out = nullptr when the
// cache is switched off, does not fit the budget or the device (the caller then lets every product build its own rotations).  widths: the distinct
// block-column widths of the batches, for the active-baby tables (matmult.go:1326-1336).  *out is the context's scratch entry "assoc.rotf" (kept for the next call).
int assoc_build_rotcache(sfg_ctx *ctx, const u64 *A_dev, int s, int in_level, int max_level, size_t nr, const std::vector<size_t> &widths, double **out) {
    *out = nullptr;
    const size_t slots = SFG_SLOTS;
    size_t jobw = 0, tailw = 0;
    const int nbr = (int)((nr + slots - 1) / slots);
    if (!ctx->cfg.assoc_cache_budget || !mac_use_dma(ctx) || sfg_rotcache_layout(ctx, s, max_level, &jobw, &tailw)) { ctx->err.clear(); return 0; }
    const size_t words = (size_t)nbr * s * jobw + tailw;
    if (words * 8 > ctx->cfg.assoc_cache_budget) return 0;
    double *buf = nullptr;
    if (sfg_scratch(ctx, "assoc.rotf", words * 8, (void **)&buf)) { ctx->err.clear(); return 0; }      // no room even without the buffers of earlier calls: per-batch rotations
    std::vector<std::vector<uint8_t>> tabs; assoc_baby_tabs(nr, widths, tabs);
    SFG_TRY(rotcache_build_rows_tab(ctx, A_dev, s, in_level, max_level, nbr, 0, nbr, &tabs, buf));
    *out = buf; return 0;
}
// The cache in the form the context multiplies with: the int8 MAC's rot tiles where every modulus runs on the matrix core (round 4: the scan's MAC leaves the
// fp64 kernel, the cache shrinks from 1.86 to 1.3 GB per block row at s = 13), else the fp64 operand rows, else nothing (every product rotates for itself).
int assoc_build_rot(sfg_ctx *ctx, const u64 *A_dev, int s, int in_level, int max_level, size_t nr, const std::vector<size_t> &widths, AssocRot &out) {
    out = AssocRot();
    if (ctx->cfg.assoc_cache_budget && ctx->cfg.assoc_i8) {
        std::vector<std::vector<uint8_t>> tabs; assoc_baby_tabs(nr, widths, tabs);
        SFG_TRY(i8_rotpre_build(ctx, A_dev, s, in_level, max_level, (int)tabs.size(), &tabs, ctx->cfg.assoc_cache_budget, "assoc.rot8", out.pre));
        if (out.pre.G) return 0;
    }
    return assoc_build_rotcache(ctx, A_dev, s, in_level, max_level, nr, widths, &out.f64);
}
void assoc_free_rot(AssocRot &r) { i8_rotpre_free(r.pre); r = AssocRot(); }        // (the buffers are the context's scratch: see i8_rotpre_free)
int assoc_product(sfg_ctx *ctx, const AssocRot &r, const uint64_t *A_dev, int s, int in_level, int max_level, const sfg_geno *g, unsigned flags, int nct, uint64_t *out) {
    if (r.pre.G) return matmul_resident_range_i8pre(ctx, r.pre, s, max_level, g, flags, 0, nct, out);
    if (r.f64) return sfg_matmul_resident_range_rc_dev(ctx, r.f64, s, max_level, g, flags, 0, nct, out);
    return sfg_matmul_resident_dev(ctx, A_dev, s, in_level, max_level, g, flags, out);
}

// out_dev: [s][out_ct_capacity][2][max_level][N]; *out_ct = sum over batches of ceil(kept / slots) (the width ConcatCipherMatrix would give).
// sum_host / sqsum_host: optional [*out_ct * slots] column sums in the reference's padded layout (dosageSum[outShift + c], assoc.go:404-405).
// One engine for both on-disk formats: a batch is a contiguous byte range of the file (.bed: nsnp * bps bytes; .pgen: the variant records of the batch, preceded by
// the LD base its first records may need), read ahead by the reader thread, copied as it is, decoded on the copy queue into the batch's int8 matrix.
// part / nparts (multi-GPU scans, mgpu.hip): this call multiplies the batches k with k % nparts == part - the reference's dispatcher hands batches to
// assoc_num_blocks_parallel workers the same way (assoc.go:360-408) - and leaves the output ciphertexts and sums of the other batches untouched; `ranges`
// (optional) receives (first output ciphertext, count) of every batch it multiplied.
enum { FMT_BED = 0, FMT_PGEN = 1 };
int assoc_stream_part(sfg_ctx *ctx, int fmt, const char *path, size_t num_sample, size_t num_snp, const uint8_t *row_filter, const uint8_t *col_filter,
                      size_t batch_snps, const uint64_t *A_dev, int s, int in_level, int max_level, unsigned flags,
                      uint64_t *out_dev, size_t out_ct_capacity, size_t *out_ct, double *sum_host, double *sqsum_host,
                      int part, int nparts, std::vector<std::pair<size_t, size_t>> *ranges) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    const char *who = fmt == FMT_BED ? "assoc_stream_bed" : "assoc_stream_pgen";
    if (!batch_snps) SFG_FAIL(ctx, "%s: bad dimensions", who);
    if (flags & SFG_TRANSPOSE) SFG_FAIL(ctx, "%s: batches are multiplied as X (samples x SNPs)", who);
    const size_t N = SFG_N, slots = SFG_SLOTS, L = (size_t)max_level;
    const bool direct = (flags & SFG_STREAM_DIRECT) != 0;                                        // bypass the page cache: what a 5 TB scan from NVMe sees
    int fd = open(path, O_RDONLY);
    if (fd < 0) SFG_FAIL(ctx, "%s: cannot open %s", who, path);                                  // os.Open panics in the reference (filestream.go:59-61)
    flags &= ~SFG_STREAM_DIRECT;
    struct stat stt; uint8_t head[12] = {0};
    if (fstat(fd, &stt) || pread(fd, head, 12, 0) < 3) { close(fd); SFG_FAIL(ctx, "%s: cannot read %s", who, path); }
    PgenIndex ix; std::vector<PgenWindow> win;
    size_t bps = 0, pitch = 0;
    if (fmt == FMT_BED) {
        if (!num_sample || !num_snp) { close(fd); SFG_FAIL(ctx, "%s: bad dimensions", who); }
        bps = (num_sample + 3) / 4; pitch = bps;
        if ((size_t)stt.st_size != 3 + num_snp * bps) { close(fd); SFG_FAIL(ctx, "%s: file holds %zu bytes, expected 3 + %zu x %zu", who, (size_t)stt.st_size, num_snp, bps); }
        if (head[0] != 0x6C || head[1] != 0x1B || head[2] != 0x01) { close(fd); SFG_FAIL(ctx, "%s: not a SNP-major PLINK .bed", who); }
    } else {
        const size_t hb = (size_t)stt.st_size >= 12 ? pgen_header_bytes(head) : 0;
        if (!hb || hb > (size_t)stt.st_size) { close(fd); SFG_FAIL(ctx, "%s: %s is not a PLINK 2 .pgen in a supported storage mode", who, path); }
        std::vector<uint8_t> hdr(hb);
        if (pread(fd, hdr.data(), hb, 0) != (ssize_t)hb) { close(fd); SFG_FAIL(ctx, "%s: cannot read the header of %s", who, path); }
        if (pgen_index(ctx, hdr.data(), hb, (size_t)stt.st_size, ix)) { close(fd); return 1; }
        num_sample = ix.ns; num_snp = ix.nv; pitch = pgen_pitch(ix);
    }
    if (direct) {                                      // the header was read through the page cache; the batches go around it
        close(fd);
        fd = open(path, O_RDONLY | O_DIRECT);
        if (fd < 0) SFG_FAIL(ctx, "%s: the file system of %s does not support O_DIRECT", who, path);
    }
    std::vector<Batch> bt_all = make_batches(col_filter, num_snp, batch_snps), bt;
    std::vector<size_t> shift_of;                       // first output ciphertext of each of THIS part's batches (positions count every batch of the file)
    if (nparts < 1 || part < 0 || part >= nparts) { close(fd); SFG_FAIL(ctx, "%s: bad part", who); }
    {
        size_t sh = 0;
        for (size_t k = 0; k < bt_all.size(); k++) {
            if ((int)(k % (size_t)nparts) == part) { bt.push_back(bt_all[k]); shift_of.push_back(sh); }
            sh += (bt_all[k].kept + SFG_SLOTS - 1) / SFG_SLOTS;
        }
        if (out_ct) *out_ct = sh;
        if (sh > out_ct_capacity) { close(fd); SFG_FAIL(ctx, "%s: output needs %zu ciphertexts per row, capacity %zu", who, sh, out_ct_capacity); }
    }
    if (ranges) ranges->clear();
    size_t max_bytes = 0, max_rows = 0, max_nsnp = 0, max_kept = 0;
    if (fmt == FMT_PGEN) win.resize(bt.size());
    for (size_t k = 0; k < bt.size(); k++) {
        Batch &b = bt[k];
        if (fmt == FMT_BED) { b.off = 3 + (off_t)(b.snp0 * bps); b.bytes = b.nsnp * bps; max_rows = std::max(max_rows, b.nsnp); }
        else {
            if (pgen_window(ctx, ix, (size_t)stt.st_size, b.snp0, b.snp0 + b.nsnp, win[k])) { close(fd); return 1; }
            b.off = (off_t)win[k].f0; b.bytes = (size_t)(win[k].f1 - win[k].f0); max_rows = std::max(max_rows, win[k].nr);
        }
        max_bytes = std::max(max_bytes, b.bytes); max_nsnp = std::max(max_nsnp, b.nsnp); max_kept = std::max(max_kept, b.kept);
    }
    if (bt.empty()) { close(fd); return 0; }
    // row map once; column maps per batch
    std::vector<int32_t> rmap_h(num_sample); size_t nr = 0;
    for (size_t i = 0; i < num_sample; i++) rmap_h[i] = (!row_filter || row_filter[i]) ? (int32_t)nr++ : -1;
    if (!nr) { close(fd); SFG_FAIL(ctx, "%s: the row filter keeps nothing", who); }
    int rc = 0;
    int32_t *rmap = nullptr, *cmap[2] = {nullptr, nullptr}; uint8_t *hb[2] = {nullptr, nullptr}, *db[2] = {nullptr, nullptr}, *rows[2] = {nullptr, nullptr}, *desc[2] = {nullptr, nullptr};
    int8_t *gb[2] = {nullptr, nullptr}; int *herr = nullptr;
    AssocRot rot;
    u64 *tmp = nullptr; hipStream_t copy = nullptr; hipEvent_t ev_h2d[2] = {nullptr, nullptr}, ev_ready[2] = {nullptr, nullptr}, ev_free[2] = {nullptr, nullptr};
    const size_t ctw = 2 * L * N, max_ct = (max_kept + slots - 1) / slots;
    // every buffer of the call is scratch of the context (device pool / pinned host pool): the next call of the scan - gWY makes four per block - finds them in place
    auto cleanup = [&]() {
        (void)hipStreamSynchronize(ctx->stream); if (copy) (void)hipStreamSynchronize(copy);
        for (int i = 0; i < 2; i++) { if (ev_h2d[i]) (void)hipEventDestroy(ev_h2d[i]); if (ev_ready[i]) (void)hipEventDestroy(ev_ready[i]); if (ev_free[i]) (void)hipEventDestroy(ev_free[i]); }
        assoc_free_rot(rot); if (copy) (void)hipStreamDestroy(copy); close(fd);
    };
#define ST_HIP(call) do { hipError_t _e = (call); if (_e != hipSuccess) { char _b[256]; snprintf(_b, sizeof _b, "%s: %s failed: %s", who, #call, hipGetErrorString(_e)); ctx->err = _b; rc = 1; } } while (0)
    const auto t_call = std::chrono::steady_clock::now();
    rc = sfg_scratch(ctx, "assoc.rmap", num_sample * sizeof(int32_t), (void **)&rmap);
    if (!rc) ST_HIP(hipMemcpy(rmap, rmap_h.data(), num_sample * sizeof(int32_t), hipMemcpyHostToDevice));
    if (!rc) rc = sfg_scratch(ctx, "assoc.tmp", (size_t)s * max_ct * ctw * 8, (void **)&tmp);
    if (!rc) ST_HIP(hipStreamCreateWithFlags(&copy, hipStreamNonBlocking));
    if (!rc && fmt == FMT_PGEN) rc = sfg_host_scratch(ctx, "assoc.herr", 2 * sizeof(int), (void **)&herr);
    for (int i = 0; i < 2 && !rc; i++) {
        const std::string sx = std::to_string(i);
        rc = sfg_host_scratch(ctx, ("assoc.hb" + sx).c_str(), max_bytes + 8192, (void **)&hb[i]);       // + the alignment slack of O_DIRECT ranges
        if (!rc) rc = sfg_scratch(ctx, ("assoc.db" + sx).c_str(), max_bytes + 16, (void **)&db[i]);
        if (!rc) rc = sfg_scratch(ctx, ("assoc.gb" + sx).c_str(), nr * max_kept, (void **)&gb[i]);
        if (!rc) rc = sfg_scratch(ctx, ("assoc.cmap" + sx).c_str(), max_nsnp * sizeof(int32_t), (void **)&cmap[i]);
        if (!rc && fmt == FMT_PGEN) { rc = sfg_scratch(ctx, ("assoc.rows" + sx).c_str(), max_rows * pitch, (void **)&rows[i]); if (!rc) rc = sfg_scratch(ctx, ("assoc.desc" + sx).c_str(), pgen_desc_bytes(max_rows), (void **)&desc[i]); }
        if (!rc) ST_HIP(hipEventCreateWithFlags(&ev_h2d[i], hipEventDisableTiming));
        if (!rc) ST_HIP(hipEventCreateWithFlags(&ev_ready[i], hipEventDisableTiming));
        if (!rc) ST_HIP(hipEventCreateWithFlags(&ev_free[i], hipEventDisableTiming));
    }
    if (rc) { cleanup(); return rc; }
#ifdef SFG_AB
    const bool trace = getenv("SFG_ASSOC_TRACE") != nullptr;
#else
    const bool trace = false;
#endif       // (debug: wall times of the cache build and of every batch's product, each synchronised)
    if (trace) fprintf(stderr, "[assoc] buffers: %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count());
    // the reader fills the first two slots while the rotation cache is built
    Reader rd; rd.fd = fd; rd.bt = &bt; rd.slot[0] = hb[0]; rd.slot[1] = hb[1]; rd.direct = direct;
    std::thread reader([&rd] { rd.run(); });
    // ---- the baby-step rotation cache of `mat`, once for all batches of the call
    {
        std::vector<size_t> widths;
        for (const Batch &b : bt) for (size_t c0 = 0; c0 < b.kept; c0 += slots) { const size_t w = std::min(slots, b.kept - c0); if (std::find(widths.begin(), widths.end(), w) == widths.end()) widths.push_back(w); }
        const auto t0 = std::chrono::steady_clock::now();
        rc = assoc_build_rot(ctx, (const u64 *)A_dev, s, in_level, max_level, nr, widths, rot);
        if (rc) { rd.release(bt.size() + 2); reader.join(); cleanup(); return rc; }
        if (trace) { (void)hipStreamSynchronize(ctx->stream); fprintf(stderr, "[assoc] rotation cache (%s): %.1f ms\n", rot.pre.G ? "int8 tiles" : rot.f64 ? "fp64 rows" : "none",
                                                                       std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count()); }
    }
    std::vector<int32_t> cmap_h(max_nsnp);
    for (size_t k = 0; k < bt.size() && !rc; k++) {
        const Batch &b = bt[k]; const int sl = (int)(k & 1);
        const size_t out_shift = shift_of[k];
        if (!rd.wait_filled(k)) { ctx->err = std::string(who) + ": " + rd.err; rc = 1; break; }
        size_t kc = 0;
        for (size_t j = 0; j < b.nsnp; j++) cmap_h[j] = (!col_filter || col_filter[b.snp0 + j]) ? (int32_t)kc++ : -1;
        // copy queue: file bytes and column map of batch k into slot sl (free once the product of batch k - 2 has run), decode into gb[sl]
        if (k >= 2) ST_HIP(hipStreamWaitEvent(copy, ev_free[sl], 0));
        if (!rc) ST_HIP(hipMemcpyAsync(db[sl], hb[sl] + rd.lead[sl], b.bytes, hipMemcpyHostToDevice, copy));
        if (!rc) ST_HIP(hipMemcpyAsync(cmap[sl], cmap_h.data(), b.nsnp * sizeof(int32_t), hipMemcpyHostToDevice, copy));
        if (!rc && fmt == FMT_PGEN) rc = pgen_upload_desc(ctx, copy, ix, win[k], desc[sl]);
        if (!rc) ST_HIP(hipEventRecord(ev_h2d[sl], copy));
        if (!rc && fmt == FMT_BED) rc = launch_bed_decode(ctx, copy, db[sl], bps, num_sample, b.nsnp, rmap, cmap[sl], gb[sl], b.kept);
        if (!rc && fmt == FMT_PGEN) {
            const int *err_dev = nullptr;
            rc = launch_pgen_decode(ctx, copy, db[sl], desc[sl], win[k].nr, ix.ns, pitch, rows[sl], &err_dev);
            if (!rc) rc = launch_bed_decode_lut(ctx, copy, rows[sl] + win[k].lead * pitch, pitch, num_sample, b.nsnp, rmap, cmap[sl], gb[sl], b.kept, 0xFF020100u);
            if (!rc) ST_HIP(hipMemcpyAsync(&herr[sl], err_dev, sizeof(int), hipMemcpyDeviceToHost, copy));
        }
        if (!rc) ST_HIP(hipEventRecord(ev_ready[sl], copy));
        if (!rc) ST_HIP(hipEventSynchronize(fmt == FMT_PGEN ? ev_ready[sl] : ev_h2d[sl]));   // the pinned slot (and cmap_h, the descriptors) may be refilled; the previous product is still running
        rd.release(k);
        if (!rc && fmt == FMT_PGEN) rc = pgen_decode_error(ctx, herr[sl]);
        if (rc) break;
        // compute queue: the product of batch k (MatMult4Stream(cps, mat, X, maxLevel, false, square, nproc), assoc.go:395), rows copied into place
        ST_HIP(hipStreamWaitEvent(ctx->stream, ev_ready[sl], 0));
        sfg_geno g; g.dev = gb[sl]; g.nrow = nr; g.ncol = b.kept; g.ld = b.kept; g.owned = false;
        const size_t nct = (b.kept + slots - 1) / slots;
        const auto tb = std::chrono::steady_clock::now();
        if (!rc) rc = assoc_product(ctx, rot, A_dev, s, in_level, max_level, &g, flags, (int)nct, (uint64_t *)tmp);
        if (trace) { const double t_enq = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb).count(); (void)hipStreamSynchronize(ctx->stream);
                     fprintf(stderr, "[assoc] batch %zu: enqueued in %.1f ms, done after %.1f ms\n", k, t_enq, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb).count()); }
        for (int i = 0; i < s && !rc; i++)
            ST_HIP(hipMemcpyAsync(out_dev + ((size_t)i * out_ct_capacity + out_shift) * ctw, tmp + (size_t)i * nct * ctw, nct * ctw * 8, hipMemcpyDeviceToDevice, ctx->stream));
        if (!rc && (sum_host || sqsum_host)) {
            if (sum_host) std::fill(sum_host + out_shift * slots, sum_host + (out_shift + nct) * slots, 0.0);
            if (sqsum_host) std::fill(sqsum_host + out_shift * slots, sqsum_host + (out_shift + nct) * slots, 0.0);
            rc = sfg_geno_colsums(ctx, &g, sum_host ? sum_host + out_shift * slots : nullptr, sqsum_host ? sqsum_host + out_shift * slots : nullptr);
        }
        if (!rc) ST_HIP(hipEventRecord(ev_free[sl], ctx->stream));
        if (ranges) ranges->push_back({out_shift, nct});
    }
#undef ST_HIP
    if (rc) rd.release(bt.size() + 2);                             // let the reader run out
    reader.join();
    const auto t_end = std::chrono::steady_clock::now();
    cleanup();
    if (trace) fprintf(stderr, "[assoc] call: %.1f ms until the last batch is enqueued, %.1f ms with the queues drained\n", std::chrono::duration<double, std::milli>(t_end - t_call).count(),
                       std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count());
    return rc;
}
extern "C" int sfg_assoc_stream_bed(sfg_ctx *ctx, const char *bed_path, size_t num_sample, size_t num_snp, const uint8_t *row_filter, const uint8_t *col_filter,
                                    size_t batch_snps, const uint64_t *A_dev, int s, int in_level, int max_level, unsigned flags,
                                    uint64_t *out_dev, size_t out_ct_capacity, size_t *out_ct, double *sum_host, double *sqsum_host) {
    ApiScope api_scope(ctx);
    return assoc_stream_part(ctx, FMT_BED, bed_path, num_sample, num_snp, row_filter, col_filter, batch_snps, A_dev, s, in_level, max_level, flags, out_dev, out_ct_capacity, out_ct,
                             sum_host, sqsum_host, 0, 1, nullptr);
}
// the same scan straight from a PLINK 2 .pgen on disk (the reference's input at config 5: 10 M SNPs per party do not fit host memory as one image); sample and
// variant counts come from the file's header
extern "C" int sfg_assoc_stream_pgen(sfg_ctx *ctx, const char *pgen_path, const uint8_t *row_filter, const uint8_t *col_filter,
                                     size_t batch_snps, const uint64_t *A_dev, int s, int in_level, int max_level, unsigned flags,
                                     uint64_t *out_dev, size_t out_ct_capacity, size_t *out_ct, double *sum_host, double *sqsum_host) {
    ApiScope api_scope(ctx);
    return assoc_stream_part(ctx, FMT_PGEN, pgen_path, 0, 0, row_filter, col_filter, batch_snps, A_dev, s, in_level, max_level, flags, out_dev, out_ct_capacity, out_ct,
                             sum_host, sqsum_host, 0, 1, nullptr);
}
