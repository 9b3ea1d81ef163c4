// common.hpp — shared host/device definitions of libsfgwas_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>
#include <map>
#include <set>
#include "../../include/sfgwas_hip.h"

typedef unsigned long long u64;
typedef unsigned __int128 u128;

constexpr int SFG_LOGN = 14;
constexpr int SFG_N = 1 << SFG_LOGN;      // ring degree (PN14QP438, gwas.go:169)
constexpr int SFG_SLOTS = SFG_N / 2;
constexpr int SFG_D = 91;                 // ceil(sqrt(8192)), matmult.go:1047
constexpr int SFG_MAXMOD = 16;
constexpr unsigned long long SFG_I8_BIG_QMAX = 0x7F7F7F7F7F80ULL;   // largest modulus whose canonical words (<= q - 1) fit six signed base-256 digits (mac_i8.hip, the NTT's six digit planes)

// per-modulus constants, device copy
struct ModConst {
    double q, qinv;        // q and 1/q as doubles
    double ninv, ninv_q;   // N^-1 mod q and (N^-1 mod q)/q  (INTT scaling)
    u64 qi;                // q as integer
};

struct RotKey {
    u64 *key_dev = nullptr;        // [beta][2][nmod][N] normal form
    uint16_t *index_dev = nullptr; // automorphism index map (N entries)
};

struct PhaseStat { double ms = 0; int launches = 0; double bytes = 0; };   // bytes = algorithmic bytes credited to the phase
struct PendingEvent { hipEvent_t e0, e1; std::string name; int launches; double bytes; };

struct sfg_geno {
    const int8_t *dev = nullptr;           // int8 [nrow][ld], or (packed) 2-bit codes: 4 columns per byte, row stride ld BYTES (multiple of 4)
    size_t nrow = 0, ncol = 0, ld = 0;
    bool owned = false;
    bool packed = false;                   // codes 0, 1, 2 = the genotype, 3 = missing (sfg_geno_pack)
    // plaintext coefficient cache (sfg_geno_set_plaintext_cache, matmul.hip): per stored block (and SFG_SQUARE) the encoder's rounded coefficient rows,
    // [8192 shifts][N/2] doubles = 512 MB, kept from the first product that touches the block; later products over it in EITHER orientation skip skew + FFT
    struct PtcEntry { double *slot; bool transposed; };
    mutable std::map<uint64_t, PtcEntry> ptc;
    mutable size_t ptc_budget = 0, ptc_used = 0;
    mutable double *ptc_arena = nullptr;   // ONE allocation of ptc_budget bytes, slots carved in order
    mutable uint32_t *ptc_perm = nullptr;  // device [8192]: shift t of the other orientation = cached row u | g << 16, image under X -> X^g (k_ntt_half3<true>)
    mutable const void *ptc_owner = nullptr;   // the context whose stream orders the fills before the hits
    mutable size_t ptc_hits = 0, ptc_fills = 0;
};

// A/B and diagnostic switches: read ONCE from the environment by sfg_ctx_create (never on the launch path)
struct SfgConfig {
    bool mac_reg = false;          // SFG_MAC_IMPL=reg      register-staged MAC kernel (mac.hip)
    bool mac_bc = true;            // SFG_MAC_IMPL=dma      the 8 x 3-tile LDS-DMA kernel (mac_dma.hip) instead of the DPP-broadcast kernel (mac_bc.hip)
    bool mac_i8 = true;            // SFG_MAC_IMPL=bc       the DPP-broadcast fp64 kernel for every modulus (round 2's MAC) instead of: small moduli on the int8 matrix core (mac_i8.hip), the 46-bit one on the DPP-broadcast kernel
    size_t i8_keep_reserve = 80ULL << 30;   // SFG_I8_KEEP_RESERVE_GB  HBM that must stay free beside the transposed copies of ALL groups of a caller's rotation cache (association scan) for the int8 MAC to take that call
    bool mac_i8_big = true;        // SFG_MAC_I8_BIG=0      the 46-bit modulus on the fp64 DPP-broadcast kernel k_mac_bc<true> (round 3's default) instead of the int8 matrix core (six digits, 36 products, eleven sums; round 4, same box: 11.14 s against 11.85 s per power iteration - mac_big 0.53 s + 0.5 s of transposition against 1.83 s).  Forced off for a ciphertext modulus above SFG_I8_BIG_QMAX
    bool mac_i8_nolds = true;      // SFG_MAC_I8_ROT=lds    int8 MAC: rot tiles of a coefficient pair staged through LDS (k_mac_i8_lds) instead of shared through the cache (measured at 100k x 1M: 3.31 s against 2.56 s per step - the barriers cost more than the re-fetches)
    double tie_band = 0x1p-50;     // distance from a rounding tie inside which the encoder's double-double value does not prove the rounding (SFG_TEST_TIE_BAND_LOG2 widens it under the test switch)
    bool test_hooks = false;       // SFG_ENABLE_TEST_HOOKS=1   sfg_ctx_encoder_inject_unsafe_for_test may be called (tests of the failure path only)
    bool mac_i8_ring = true;       // SFG_MAC_I8_ROT=cache    int8 MAC without the LDS prefetch ring (k_mac_i8: operands straight from global memory, rot tiles shared through the L1)
    bool stage_pack = false;       // SFG_MAC_I8_STAGE=1    int8 MAC: streamed transposition (StagePack, kernels.hpp) instead of the full plaintext panel and a transposition pass per MAC launch.  Built, bit-exact, and measured SLOWER at 100k x 1M (14.2 s against 12.5 s per step on one box: the small per-batch transposition launches run at 1.5 TB/s and slow the encode kernels they share the chip with)
    int stage_giants = 11;         // SFG_STAGE_GIANTS=n     giant steps per batch of the streamed transposition (16: a batch completes whole 16-column tiles)
    bool stage_same_queue = false; // SFG_STAGE_SAMEQ=1      the batch transposition on the product's own queue (behind its NTT) instead of the encode queue
    int mac_i8_waves = 12;         // SFG_MAC_I8_WAVES=6     k_mac_i8_ring with six waves per coefficient pair (a wave = both coefficients x 16 columns) instead of twelve (one coefficient each)
    int mac_i8_diag = 0;           // SFG_MAC_I8_DIAG=1 / 2   timing diagnostics of k_mac_i8_ring, results INVALID: 1 = one MFMA per rot tile, 2 = no DMA after the prologue
    bool mac_i8_wg1 = false;       // SFG_MAC_I8_WG=1       int8 MAC diagnostic: one column wave per workgroup (no cache shared between the column waves of a coefficient pair); for the PMC re-fetch measurement
    int mac_wc = 1;                // SFG_MAC_WC            column waves per small-modulus MAC workgroup
    int mm_group = 8;              // SFG_MM_GROUP          block rows per MAC launch
    bool mm_group_auto = true;     //                       (unset) 16 block rows per launch when the plaintext panel and the rotation operands of such a group fit the free HBM, else 8
    size_t acc_budget = 24ULL << 30;   // SFG_MM_ACC_BUDGET_MB
    bool no_overlap = true;        // SFG_MM_OVERLAP=1      two queues: the key switching of the next block-row group / the giant-step alignment of the last column pass beside the encode + MAC.  Off since round 4 (single queue: 10.87 s against 10.91 s per step at 100k x 1M, 3.39 against 3.41 s at 50k x 500k - the fp64-issue-bound kernels only slow each other down beside the HBM-bound ones); SFG_MM_NO_OVERLAP=1 is still accepted
    bool no_enc_overlap = true;    // SFG_MM_ENC_OVERLAP=1  the encode of MAC launch k + 1 on a third queue beside the transposition + MAC of launch k (two plaintext panels).  Built and measured at
                                   // 100k x 1M: 12.24 s against 12.20 s - the kernels then share the machine in time, not in space: a MAC workgroup (6 waves x 240 VGPRs) leaves no SIMD
                                   // with the 128 VGPRs a plaintext-NTT wave needs, so an encode workgroup cannot be resident beside it.  Off until the MAC leaves that room.
    bool ntt_fwd_full = false;     // SFG_NTT_FWD_IMPL=full   one 512-thread workgroup per row for the general forward NTT (instead of two half-row workgroups)
    bool ntt_half_full = false;    // SFG_NTT_HALF_IMPL=full
    bool upload_blocking = false;  // SFG_UPLOAD_BLOCKING   blocking pointer-table uploads (rocprofv3 --pmc)
    size_t ksw_budget = 4ULL << 30; // SFG_KSW_BUDGET_MB      key-switch scratch per input group / job chunk: more jobs per chunk = more reuse of a key (64 MB: +45 %, 1.5 GB: +2 %, 12 GB: -2 %)
    std::string test_scratch_oom;  // SFG_TEST_SCRATCH_OOM=name:n  (test switch only) the n-th request of scratch buffer `name` inside a top-level call behaves as if the device were full: the eviction path of sfg_scratch runs
    int i8_mover = 0;              // SFG_I8_MOVER=n (A/B build)  n workgroups of the plaintext transposition in its mover form (i8_move.hpp) instead of the pass k_i8_pack_pt_digits: 1280 is 3 % faster alone, nothing in a product (profiles/r06_mover_ubench.txt)
    int i8_mover_depth = 3;        // SFG_I8_MOVER_DEPTH     units (32 KiB) a mover workgroup keeps in flight + 1
    bool pt_compact = true;        // panel rows of an all-int8 product hold only their digit planes (208 KiB per plaintext instead of 320 KiB at L = 5).  SFG_PT_COMPACT=0 in the A/B build
    bool pt_kmajor = true;         // the compact panel K-major: [column][plane][128-byte coefficient block][k][128 B] (2 KiB source runs for the transposition).  SFG_PT_KMAJOR=0 in the A/B build
    int pt_ride = 192;             // mover workgroups of the riding transposition per plaintext-NTT launch (kernels.hpp PtRide; 0 = the transposition pass before every MAC launch).  SFG_PT_RIDE in the A/B build
    int i8_mover_depth_ride = 1, i8_mover_nt_ride = 1;     // (A/B build: SFG_PT_RIDE_DEPTH, SFG_PT_RIDE_NT)
    int enc_batch = 2048;          // SFG_ENC_BATCH          diagonals per FFT / plaintext-NTT launch pair: 128 MB of coefficient rows stay cache resident between the two now that the NTT's digit planes leave by streaming stores (round 5: 2048 -3 % of a 50k x 500k step against 1024, 3072 the same, 4096 worse; with plain stores 1024 was best)
    bool mac_plain_pt = false;     // SFG_MAC_PT=plain      plaintext panel as plain u64 words (A/B of the packed-limb panel format)
    // CU partitioning experiments (round 5): restrict a queue of the context to a set of compute units, "lo-hi[,lo-hi...]" over the bits of hipExtStreamCreateWithCUMask
    // (bit i = CU i of the device's enumeration).  Empty = all CUs (the default).  A stream installed by sfg_ctx_set_stream is the caller's and keeps its own mask.
    std::string cu_main, cu_enc, cu_aux;   // SFG_CU_MAIN / SFG_CU_ENC / SFG_CU_AUX
    bool assoc_i8 = true;                   // SFG_ASSOC_I8=0           association scan: keep the rotation cache as fp64 operand rows (round 3) instead of the int8 MAC's rot tiles
    size_t assoc_cache_budget = 160ULL << 30;   // SFG_ASSOC_ROTCACHE_MB   largest baby-step rotation cache sfg_assoc_stream_bed keeps across the batches of a call (0: rebuild per batch, the A/B switch)
};

// Immutable after setup, shared by a context and its forks (sfg_ctx_fork): ring tables, encoder tables, key material.
struct SfgShared {
    int device = 0;
    int logN = SFG_LOGN, N = SFG_N, nq = 0, np = 0, nmod = 0, beta = 0;
    double scale = 0;
    u64 q[SFG_MAXMOD] = {0}, psi[SFG_MAXMOD] = {0};
    double *tw_fwd = nullptr;    // [nmod][N] w, index m+i as in the CT loop (psi^bitrev)
    double *tw_inv = nullptr;    // [nmod][N] w for psi^-bitrev
    double2 *pack_fwd = nullptr; // [nmod][256][8][64] late-stage (t <= 8) twiddles packed for coalesced per-wave loads
    double2 *pack_inv = nullptr;
    ModConst *modc = nullptr;    // [nmod]
    ModConst modc_host[SFG_MAXMOD];
    void *enc_tables = nullptr;  // encoder tables (double-double twiddles), see encode.hip
    void *zeros_dev = nullptr;   // 256 B of zeros (DMA source for padded k-steps)
    u64 *sk_dev = nullptr;       // secret-key shard [nq][N], NTT domain, canonical (sfg_ctx_load_secret_key; collective bootstrap shares)
    std::map<u64, RotKey> rotkeys;      // written only by sfg_ctx_load_rotkey / _relinkey (setup time), read by every fork
    SfgConfig cfg;
    int refs = 1;                // the creating context + live forks
};

// A context = the shared part + ONE caller's execution state (streams, scratch, staging ring, timers, error string).
// Concurrent callers (assoc.go:360-408 runs assoc_num_blocks_parallel MatMult4Stream calls at once) each use their own
// fork; nothing below is touched by another thread.
struct sfg_ctx {
    SfgShared *sh = nullptr;
    bool is_fork = false;
    // mirrors of the shared scalars / table pointers (read-only; filled by ctx_bind_shared)
    int device = 0;
    int logN = SFG_LOGN, N = SFG_N, nq = 0, np = 0, nmod = 0, beta = 0;
    double scale = 0;
    u64 q[SFG_MAXMOD] = {0}, psi[SFG_MAXMOD] = {0};
    double *tw_fwd = nullptr, *tw_inv = nullptr;
    double2 *pack_fwd = nullptr, *pack_inv = nullptr;
    ModConst *modc = nullptr;
    ModConst modc_host[SFG_MAXMOD];
    SfgConfig cfg;
    // int8 MAC (mac_i8.hip): generation of the fp64 rot operands - bumped by whoever rewrites a rotation cache - and the two transposed copies keyed by it
    unsigned ntt_plain_seq = 0;      // sampling counter of the panel-NTT phase timer (encode.hip)
    // A copy is recycled when its generation is stale (a product's groups), kept while it is current (a caller's rotation cache multiplied batch after batch:
    // one copy per group of the cache, as many as the free HBM takes).
    struct I8Slot { const void *src = nullptr; u64 sig[8] = {0, 0, 0, 0, 0, 0, 0, 0}; u64 last_use = 0; };
    static constexpr int I8_SLOTS = 16;
    u64 i8_gen = 1, i8_clock = 0; I8Slot i8_slot[2][I8_SLOTS];   // [0] 35-bit moduli, [1] the 46-bit one
    hipStream_t own_stream = nullptr, stream = nullptr;
    hipStream_t user_stream = nullptr;   // installed by sfg_ctx_set_stream (nullptr = own_stream is the main queue)
    hipStream_t aux_stream = nullptr;    // second queue: key switching of the next group / previous column pass runs beside encode + MAC
    // pinned host ring for small stream-ordered uploads (pointer tables): no blocking copies on the launch path
    unsigned char *pin = nullptr; size_t pin_bytes = 0, pin_head = 0;
    hipStream_t enc_stream = nullptr;    // third queue: the encode (skew, FFT, NTT: fp64-issue bound) of MAC launch k + 1 beside the HBM-bound transposition + int8 MAC of launch k
    hipEvent_t ev_enc[4] = {nullptr, nullptr, nullptr, nullptr};    // [0,1]: panel buffer encoded; [2,3]: panel buffer consumed by its MAC
    hipEvent_t ev_pipe[4] = {nullptr, nullptr, nullptr, nullptr};   // [0,1]: rotation cache of group parity ready; [2,3]: finalize of column pass parity done
    std::vector<hipEvent_t> ev_pool; size_t ev_next = 0;     // ordering events (no timing), reused round-robin
    std::map<int, void *> ksw_cache;   // per-level key-switch constants (device), rotate.hip
    // scratch
    void *ws = nullptr; size_t ws_bytes = 0;
    std::map<std::string, std::pair<void *, size_t>> pool;   // named grow-only device scratch (sfg_scratch), freed with the context
    std::map<std::string, std::pair<void *, size_t>> host_pool;   // named grow-only PINNED host scratch (sfg_host_scratch): the streamed scan's two file slots; freed with the context and by sfg_ctx_release_scratch
    std::map<std::string, unsigned long long> pool_epoch;    // the top-level call (ApiScope) that last asked for the buffer: when the device is full, buffers no call in progress uses are given back
    unsigned long long api_epoch = 0; int api_depth = 0;
    int scratch_oom_seen = 0;                               // requests of cfg.test_scratch_oom's buffer so far (test switch)
    std::vector<PendingEvent> pending;                      // phase timers not yet read back (resolved by sfg_phases_resolve)
    std::string err;
    std::map<std::string, PhaseStat> phases;
    int sp_shape = -1;                      // block rows per group the streamed tile buffers (mi8.Bs / mi8.Bb) were last cleared for
    std::set<const sfg_geno *> ptc_genos;   // matrices whose plaintext coefficient cache this context owns (dropped when an unprovable encoder rounding is reported / reset)
    bool test_hooks = false;                // SFG_ENABLE_TEST_HOOKS=1 at context creation: the failure-path test hook may be used
    void *tie_count_dev = nullptr;          // counters (+ a third: coefficients inside the band whose rounding the exact re-derivation proved): encoder coefficients within 2^-40 of a rounding tie (audit) and within 2^-50 (sticky failure, sfg_encoder_check)
    hipStream_t main_stream() const { return user_stream ? user_stream : own_stream; }
    std::map<u64, RotKey> &rotkeys() { return sh->rotkeys; }
    const std::map<u64, RotKey> &rotkeys() const { return sh->rotkeys; }
    void *enc_tables() const { return sh->enc_tables; }
    void *zeros_dev() const { return sh->zeros_dev; }
};

// synchronise every queue of the context (before freeing / reusing memory either queue may still read)
int sfg_sync_all(sfg_ctx *ctx);
// after a sync: non-zero while an encoder coefficient too close to a rounding tie is outstanding (encode.hip)
int sfg_encoder_check(sfg_ctx *ctx);
void sfg_ptc_detach_all(sfg_ctx *ctx);         // matmul.hip: release every plaintext coefficient cache this context owns and clear the owner of its matrices (context destruction)
void sfg_ptc_invalidate_all(sfg_ctx *ctx);     // matmul.hip: forget every cached coefficient row of the matrices this context owns a cache of (arenas kept)

extern thread_local std::string g_create_error;

#define SFG_FAIL(ctx, ...) do { char _b[512]; snprintf(_b, sizeof _b, __VA_ARGS__); (ctx)->err = _b; return 1; } while (0)
// (the A/B build names the failing call; the product names file and line only - its binary carries no expression text)
#ifdef SFG_AB
#define SFG_HIP(ctx, call) do { hipError_t _e = (call); if (_e != hipSuccess) { char _b[512]; \
    snprintf(_b, sizeof _b, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); (ctx)->err = _b; return 1; } } while (0)
#else
#define SFG_HIP(ctx, call) do { hipError_t _e = (call); if (_e != hipSuccess) { char _b[512]; \
    snprintf(_b, sizeof _b, "HIP call failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); (ctx)->err = _b; return 1; } } while (0)
#endif
#define SFG_TRY(expr) do { int _rc = (expr); if (_rc) return _rc; } while (0)

// workspace (grow-only scratch owned by the context)
int sfg_ws_reserve(sfg_ctx *ctx, size_t bytes);

// host modular helpers
static inline u64 h_mulmod(u64 a, u64 b, u64 q) { return (u64)(((u128)a * b) % q); }
static inline u64 h_powmod(u64 a, u64 e, u64 q) { u64 r = 1 % q; a %= q; while (e) { if (e & 1) r = h_mulmod(r, a, q); a = h_mulmod(a, a, q); e >>= 1; } return r; }
static inline u64 h_invmod(u64 a, u64 q) { return h_powmod(a, q - 2, q); }
static inline uint32_t h_brev(uint32_t x, int bits) { uint32_t r = 0; for (int i = 0; i < bits; i++) { r = (r << 1) | (x & 1); x >>= 1; } return r; }

// stream-ordered upload of a small host table: staged in the pinned ring, copied with hipMemcpyAsync on ctx->stream
int sfg_upload_small(sfg_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
// A top-level product call in progress (nested entry points share the outermost scope): sfg_scratch may evict buffers that were last requested by EARLIER calls
struct ApiScope {
    sfg_ctx *c;
    explicit ApiScope(sfg_ctx *c_) : c(c_) { if (c->api_depth++ == 0) c->api_epoch++; }
    ~ApiScope() { c->api_depth--; }
};
// run the enclosed launches on the auxiliary stream (everything launches on ctx->stream)
struct AuxScope {
    sfg_ctx *c; hipStream_t saved;
    explicit AuxScope(sfg_ctx *c_, bool on = true) : c(c_), saved(c_->stream) { if (on) c->stream = c->aux_stream; }
    ~AuxScope() { c->stream = saved; }
};
// order `waiter` after everything enqueued so far on `signaller` (no host wait)
int sfg_stream_after(sfg_ctx *ctx, hipStream_t waiter, hipStream_t signaller);
// named grow-only scratch buffers owned by the context (avoids hipMalloc/hipFree of multi-GB buffers per call)
int sfg_scratch(sfg_ctx *ctx, const char *name, size_t bytes, void **out);
int sfg_host_scratch(sfg_ctx *ctx, const char *name, size_t bytes, void **out);     // pinned host memory, same keeping rules
// reads back all pending phase events (one stream sync); called by the phase query functions and at API exits
void sfg_phases_resolve(sfg_ctx *ctx);

// phase timing helper: HIP events around a region on the ctx stream, WITHOUT a host sync (resolved lazily)
struct PhaseTimer {
    sfg_ctx *ctx; const char *name; hipEvent_t e0, e1; bool on;
    PhaseTimer(sfg_ctx *c, const char *n, bool enable = true) : ctx(c), name(n), on(enable) {
        if (on) { (void)hipEventCreate(&e0); (void)hipEventCreate(&e1); (void)hipEventRecord(e0, ctx->stream); }
    }
    void stop(int launches = 1, double bytes = 0) {
        if (!on) return;
        (void)hipEventRecord(e1, ctx->stream);
        ctx->pending.push_back(PendingEvent{e0, e1, name, launches, bytes});
        on = false;
        if (ctx->pending.size() > 4096) sfg_phases_resolve(ctx);
    }
    ~PhaseTimer() { stop(0); }
};

// Packed-limb plaintext word of a canonical residue w = x0 + x1 2^12 + x2 2^24 < 2^36 (the broadcast MAC's panel format): the three 12-bit limbs in
// 16-bit fields.  Framed by a zero byte above and below, a field IS the high dword of the SUBNORMAL double x * 2^-1034 (exponent field 0, x in mantissa
// bits 51..40), so the MAC turns a limb into an FMA operand with one v_perm_b32 and nothing to undo: products with the integer rot operand are exact
// multiples of 2^-1034, sums stay exact below 2^53 * 2^-1034, and two multiplications by 2^517 bring a sum back (fp64 subnormals run at full rate:
// tools/ubench_dpp.hip).  Zero is the all-zero word.
constexpr u64 PACKED_ZERO = 0ULL;
#ifdef __HIPCC__
__device__ __host__ __forceinline__ u64 pack_limbs(u64 w) {
    return (w & 0xFFF) | (((w >> 12) & 0xFFF) << 16) | ((w >> 24) << 32);
}
// ---------------------------------------------------------------- device arithmetic
// All ring arithmetic on the device is done on exact integers held in fp64 registers (|x| < 2^53):
// measured on gfx950 v_fma_f64 and v_mad_u64_u32 issue at the same rate (profiles/r01_ubench_*.txt),
// and the fp64 form needs no carry chains.

// x*w mod q, result in (-q, q) (not canonical). Requires |x|*w < 2^105, |x| < 2^51, wq = w/q (rounded).
__device__ __forceinline__ double mulmod_lazy(double x, double w, double wq, double q) {
    double qh = __builtin_rint(x * wq);
    double h = x * w;
    double l = __builtin_fma(x, w, -h);
    double r = __builtin_fma(-qh, q, h);
    return r + l;
}
// The same product with the quotient estimated from the rounded high part: needs only 1/q, not w/q, so butterflies that take a fresh twiddle
// spend no multiply on w * (1/q).  |x * w / q| < 2^41 keeps the estimate within 1/2 + 2^-11 of the true quotient: result in (-q, q).
__device__ __forceinline__ double mulmod_lazy_q(double x, double w, double q, double qinv) {
    double h = x * w;
    double qh = __builtin_rint(h * qinv);
    double l = __builtin_fma(x, w, -h);
    double r = __builtin_fma(-qh, q, h);
    return r + l;
}
// canonical representative in [0, q) of an integer-valued double |x| < 2^51.
// y = fl(x * fl(1/q)) is within |x/q| * 2^-52 (1 + 2^-53) of x/q.  Write x = k q + e, 0 <= e < q: for e >= 1 both e/q and (q - e)/q exceed
// that error (e >= 1 > |x| 2^-51), so floor(y) = k; for e = 0, y = k (1 + d) may fall just below k, floor(y) = k - 1 and the exact
// remainder fma(-floor(y), q, x) is q.  One equality test is therefore the whole fix-up.
__device__ __forceinline__ double canon(double x, double q, double qinv) {
    const double r = __builtin_fma(-__builtin_floor(x * qinv), q, x);
    return r == q ? 0.0 : r;
}
// partial reduction to (-q, q): cheap, for lazy sums that would otherwise grow
// the same without the fix-up: a representative in [0, q] (q itself only for multiples of q).  Enough wherever the value is an operand of further
// lazy arithmetic and merely has to be congruent, non-negative and <= q - e.g. the packed-limb plaintext words the MAC reads (q < 2^36 fits the limbs).
__device__ __forceinline__ double canon_le(double x, double q, double qinv) {
    return __builtin_fma(-__builtin_floor(x * qinv), q, x);
}
__device__ __forceinline__ double pred(double x, double q, double qinv) {
    double qh = __builtin_rint(x * qinv);
    return __builtin_fma(-qh, q, x);
}
__device__ __forceinline__ double u64_to_f64(u64 x) {            // exact for x < 2^52
    return __longlong_as_double((long long)(x | 0x4330000000000000ULL)) - 4503599627370496.0;
}
__device__ __forceinline__ u64 f64_to_u64(double x) {            // exact for integer 0 <= x < 2^52
    return (u64)__double_as_longlong(x + 4503599627370496.0) & 0x000FFFFFFFFFFFFFULL;
}
// 4 x 4 byte transpose with v_perm_b32: o[t] = {w0.b_t, w1.b_t, w2.b_t, w3.b_t}
__device__ __forceinline__ void bytes_tr4(unsigned w0, unsigned w1, unsigned w2, unsigned w3, unsigned (&o)[4]) {
    const unsigned p0 = __builtin_amdgcn_perm(w1, w0, 0x05010400u), p1 = __builtin_amdgcn_perm(w1, w0, 0x07030602u);
    const unsigned q0 = __builtin_amdgcn_perm(w3, w2, 0x05010400u), q1 = __builtin_amdgcn_perm(w3, w2, 0x07030602u);
    o[0] = __builtin_amdgcn_perm(q0, p0, 0x05040100u); o[1] = __builtin_amdgcn_perm(q0, p0, 0x07060302u);
    o[2] = __builtin_amdgcn_perm(q1, p1, 0x05040100u); o[3] = __builtin_amdgcn_perm(q1, p1, 0x07060302u);
}
// sum over 4 packed int8 of int8(x*x) (Go's uint64(row[j]*row[j]) / float64(int8(x*x)): the square wraps in int8): the dot product of the dword with
// itself when it is < 128 - then no single square reached 128 - else byte by byte
__device__ __forceinline__ int sq_sum4_i8(int w) {
    const int sq = __builtin_amdgcn_sdot4(w, w, 0, false);
    if (sq < 128) return sq;
    int s = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { const int x = (int)(int8_t)(w >> (8 * k)); s += (int)(int8_t)(x * x); }
    return s;
}
// packed-limb word (pack_limbs) of an integer-valued double 0 <= x < 2^36 straight from the bits of x + 2^52: 1 fp add + 4 integer ops
__device__ __forceinline__ u64 pack_limbs_f64(double x) {
    const u64 b = (u64)__double_as_longlong(x + 4503599627370496.0);
    const unsigned lo = (unsigned)b, hi = (unsigned)(b >> 32);
    const unsigned x01 = ((lo << 4) & 0x0FFF0000u) | (lo & 0xFFFu);
    const unsigned x2 = __builtin_amdgcn_alignbit(hi, lo, 24) & 0xFFFu;
    return ((u64)x2 << 32) | x01;
}
__device__ __forceinline__ u64 d_mulmod_u64(u64 a, u64 b, u64 q) { // generic (slow) path for setup kernels
    u64 hi = __umul64hi(a, b), lo = a * b;
    // 128-by-64 division by shift-subtract (setup only)
    u64 r = hi % q;
    for (int i = 63; i >= 0; i--) { u64 top = r >> 63; r = (r << 1) | ((lo >> i) & 1); if (top || r >= q) r -= q; }
    return r;
}
__device__ __forceinline__ u64 splitmix_at(u64 seed, u64 idx) {   // counter-mode splitmix64 (element idx, 0-based)
    u64 z = seed + 0x9E3779B97F4A7C15ULL * (idx + 1);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
#endif
