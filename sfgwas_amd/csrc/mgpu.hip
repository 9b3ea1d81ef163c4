// mgpu.hip — SURVEY §8e as a PRODUCT entry point: one party's local products on the G GPUs of a node, behind the C-ABI (sfg_mgpu_*, include/sfgwas_hip.h).
//
// The reference runs ONE OS process per party (run_example.sh:1-12) and calls MatMult4StreamCompute twice per power iteration (gwas/pca.go:344,352) and
// MatMult4Stream once per SNP batch of the association scan (gwas/assoc.go:360-408).  A Go party process therefore needs all of a node's GPUs from ONE process:
// sfg_mgpu_create makes one context per device and one RCCL communicator set (ncclCommInitAll); every product call runs one host thread per device.  The same
// engine serves one-process-per-GPU launchers (bench.py under torch.distributed.run): sfg_mgpu_create_rank joins a world with a 128-byte id carried by the caller.
//
// Partitioning (the arithmetic of sfgwas_amd/sharding.py restated in C): the genotype matrix X (n_ind x m_snp) is split by blocks of 8192 SNP columns.
//   Q  * X    (kp x n_ind)(n_ind x m_snp): rank r owns output block columns [blk0, blk1).  No data-path collective: every rank key-switches the baby-step rotation
//             cache of all inputs itself (72 ms at 100k x 1M; an all-gather of a sharded build would move 25 GB into every rank).
//   Q' * X^T  (kp x m_snp)(m_snp x n_ind): contraction over the rank's SNP blocks.  Key switching is not bit-linear, so the partial sums are combined BEFORE the
//             giant-step rotations: per output block column j, the canonical uint64 accumulators [giant][i][2][L][N] (padded to world * ceil(91 / world) giant slots)
//             are reduce-scattered over the giant axis on a second queue while column j + 1 is being multiplied (two column buffers); then every rank reduces mod q,
//             aligns ITS giant steps (sfg_matmul_finalize_slots_dev), and the aligned partial outputs are all-reduced and reduced once more.
//             Sums of `world` canonical residues stay below world * 2^47: no overflow in uint64.
//
// Transports: "rccl" (ncclReduceScatter / ncclAllReduce on uint64 over xGMI; the library resolves librccl at run time with dlopen so that single-GPU users need
// no RCCL and a process that already holds a copy - PyTorch's - shares it) and "direct" (single process only: a rank sums its slice straight out of its peers'
// buffers with a kernel - peer access over xGMI, or plain loads when several ranks share one device, which RCCL refuses: that is how world sizes 2 and 3 are
// tested on a one-GPU box; host-side rendezvous, no overlap).  Integer sums: both give identical words.
#include "common.hpp"
#include "kernels.hpp"
#include <rccl/rccl.h>          // types and prototypes only; the functions are resolved at run time (Rccl below)
#include <dlfcn.h>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>

namespace {

struct Rccl {
    void *so = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclReduceScatter) ReduceScatter = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommAbort) CommAbort = nullptr;          // optional (older builds): without it a failure after the agreement point cannot release the peers
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    std::string load(const char *configured = nullptr) {
        if (so) return "";
        const char *names[] = {getenv("SFG_RCCL_LIB"), configured, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) { if (!n || !*n) continue; so = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (so) break; }
        if (!so) return std::string("RCCL not found (dlopen librccl.so.1): ") + (dlerror() ? dlerror() : "?") + " - name the library (sfg_config.rccl_lib), or use the direct transport in a single process";
#define SFG_RCCL_SYM(f) f = (decltype(f))dlsym(so, "nccl" #f); if (!f) return std::string("RCCL symbol missing: nccl" #f)
        SFG_RCCL_SYM(GetUniqueId); SFG_RCCL_SYM(CommInitAll); SFG_RCCL_SYM(CommInitRank); SFG_RCCL_SYM(CommDestroy); SFG_RCCL_SYM(ReduceScatter); SFG_RCCL_SYM(AllReduce);
        SFG_RCCL_SYM(GetErrorString);
#undef SFG_RCCL_SYM
        CommAbort = (decltype(CommAbort))dlsym(so, "ncclCommAbort");
        CommCount = (decltype(CommCount))dlsym(so, "ncclCommCount");
        CommUserRank = (decltype(CommUserRank))dlsym(so, "ncclCommUserRank");
        return "";
    }
};
Rccl g_rccl; std::mutex g_rccl_mu;          // (function pointers of a shared object: the one process-wide datum of the library, written once under the lock)

struct MgRank {
    sfg_ctx *ctx = nullptr;
    int rank = 0, device = 0;
    ncclComm_t comm = nullptr;
    hipStream_t coll = nullptr;             // the collectives' queue (RCCL kernels beside the MAC of the next column)
    bool own_coll = false;                  // coll was created by the engine (else it is the context's encode queue)
    hipStream_t spare = nullptr;            // SFG_MGPU_COLL_QUEUE=spare (diagnostics)
    hipEvent_t ev_acc[2] = {nullptr, nullptr}, ev_rs[2] = {nullptr, nullptr}, ev_c = nullptr;
    uint64_t *status_dev = nullptr, *status_host = nullptr;      // the agreement word of a call (coll_agree): made with the rank, so that agreeing never allocates
    std::string err;
};

// host-side meeting point of the rank threads of the direct transport
struct Rendezvous {
    std::mutex m; std::condition_variable cv; int n = 1, count = 0; unsigned long gen = 0; std::atomic<bool> failed{false};
    const void *ptr[64];
    bool barrier() {       // false: a peer has failed (nobody will arrive)
        std::unique_lock<std::mutex> lk(m);
        if (failed.load()) return false;
        const unsigned long g = gen;
        if (++count == n) { count = 0; gen++; cv.notify_all(); return true; }
        cv.wait(lk, [&] { return gen != g || failed.load(); });
        return gen != g;
    }
    void fail() { std::unique_lock<std::mutex> lk(m); failed.store(true); cv.notify_all(); }
};

// out[x] = sum_p src[p][x] over n peers' slices (uint64, wraps never: world * 2^47)
struct PeerPtrs { const u64 *p[64]; };
__global__ void __launch_bounds__(256) k_sum_peers(u64 *out, PeerPtrs src, int n, size_t count) {
    for (size_t x = (size_t)blockIdx.x * 256 + threadIdx.x; x < count; x += (size_t)gridDim.x * 256) {
        u64 s = 0;
        for (int p = 0; p < n; p++) s += src.p[p][x];
        out[x] = s;
    }
}

}  // namespace

struct sfg_mgpu {
    int world = 1;
    bool single_process = true, direct = false, force_coll = false;
    bool solo = false;                      // SFG_MGPU_SOLO=r/w: TIMING ONLY - this process computes the share of rank r of a w-rank world on one GPU, every exchange replaced by a
                                            // local copy of the rank's own slice (the outputs are not a product): per-rank phase times of world sizes a one-GPU box cannot run
    size_t cache_budget = 72ULL << 30;      // sfg_config.mgpu_cache_bytes / SFG_MGPU_CACHE_GB: a rank's own Q'X^T rotation cache up to this size -> per-column pipelined reduce-scatter
    std::string rccl_lib;                   // sfg_config.rccl_lib
    std::vector<MgRank> r;                  // local ranks
    bool broken = false;                    // a rank failed after a call's agreement point and its communicator was aborted: the engine refuses further exchanges
    Rendezvous rv;
    std::string err;
};
struct sfg_mgeno {
    size_t nrow = 0, ncol = 0;              // of the GLOBAL matrix
    std::vector<sfg_geno *> shard;          // per local rank: its SNP-block window [col0, col1) as a resident matrix (nullptr: the rank owns no block)
    std::vector<void *> owned;              // per local rank: device buffer behind a non-owning handle (nullptr: the handle owns its memory)
    std::vector<size_t> blk0, blk1;
};

thread_local std::string g_mgpu_create_error;
#define MG_FAIL(mg, ...) do { char _b[640]; snprintf(_b, sizeof _b, __VA_ARGS__); (mg)->err = _b; return 1; } while (0)
#define R_FAIL(R, ...) do { char _b[640]; snprintf(_b, sizeof _b, __VA_ARGS__); (R).err = _b; return 1; } while (0)
#ifdef SFG_AB
#define R_HIP(R, call) do { hipError_t _e = (call); if (_e != hipSuccess) R_FAIL(R, "%s failed: %s (%s:%d)", #call, hipGetErrorString(_e), __FILE__, __LINE__); } while (0)
#define R_CTX(R, call) do { if ((call)) { (R).err = std::string(#call).substr(0, std::string(#call).find('(')) + ": " + (R).ctx->err; return 1; } } while (0)
#define R_NCCL(R, call) do { ncclResult_t _e = (call); if (_e != ncclSuccess) R_FAIL(R, "%s failed: %s", #call, g_rccl.GetErrorString(_e)); } while (0)
#else         // (the product's binary carries no expression text: file and line, and the context's own message)
#define R_HIP(R, call) do { hipError_t _e = (call); if (_e != hipSuccess) R_FAIL(R, "HIP call failed: %s (%s:%d)", hipGetErrorString(_e), __FILE__, __LINE__); } while (0)
#define R_CTX(R, call) do { if ((call)) { (R).err = "mgpu.hip:" + std::to_string(__LINE__) + ": " + (R).ctx->err; return 1; } } while (0)
#define R_NCCL(R, call) do { ncclResult_t _e = (call); if (_e != ncclSuccess) R_FAIL(R, "RCCL call failed: %s (mgpu.hip:%d)", g_rccl.GetErrorString(_e), __LINE__); } while (0)
#endif

// one host thread per local rank (the library's rule: one thread drives a context at a time); first failure wins
template <class F> static int run_ranks(sfg_mgpu *mg, F &&fn) {
    const int n = (int)mg->r.size();
    std::vector<int> rc((size_t)n, 0);
    for (auto &R : mg->r) R.err.clear();
    mg->rv.failed.store(false);
    auto body = [&](int i) { rc[(size_t)i] = fn(mg->r[(size_t)i], i); if (rc[(size_t)i]) mg->rv.fail(); };
    if (n == 1) body(0);
    else { std::vector<std::thread> th; for (int i = 0; i < n; i++) th.emplace_back(body, i); for (auto &t : th) t.join(); }
    for (int i = 0; i < n; i++) if (rc[(size_t)i] && !mg->r[(size_t)i].err.empty() && mg->r[(size_t)i].err != "a peer rank failed") {
        mg->err = "rank " + std::to_string(mg->r[(size_t)i].rank) + " (device " + std::to_string(mg->r[(size_t)i].device) + "): " + mg->r[(size_t)i].err; return 1; }
    for (int i = 0; i < n; i++) if (rc[(size_t)i]) { mg->err = "rank " + std::to_string(mg->r[(size_t)i].rank) + ": " + (mg->r[(size_t)i].err.empty() ? mg->r[(size_t)i].ctx->err : mg->r[(size_t)i].err); return 1; }
    return 0;
}

// One top-level engine call for the scratch pools of EVERY rank (the ApiScope of common.hpp across the rank threads of a call: the phases of a call run in separate
// run_ranks passes, and a buffer requested in the first one - mg.Ain, mg.Oout - must not look like an earlier call's when a later phase runs out of device memory).
// Contexts are driven by one thread at a time; run_ranks joins its threads, so the calling thread may touch the counters between passes.
struct MgApiScope {
    sfg_mgpu *mg;
    explicit MgApiScope(sfg_mgpu *m) : mg(m) { for (auto &R : mg->r) if (R.ctx && R.ctx->api_depth++ == 0) R.ctx->api_epoch++; }
    ~MgApiScope() { for (auto &R : mg->r) if (R.ctx) R.ctx->api_depth--; }
};
#define MG_NEED(mg, cond, what) do { if (!(cond)) { if (mg) MG_FAIL(mg, "%s: %s", __func__, what); g_mgpu_create_error = std::string(__func__) + ": " + what; return 1; } } while (0)

extern "C" int sfg_mgpu_shard(int world, size_t ncol, int rank, size_t *blk0, size_t *blk1, size_t *col0, size_t *col1) {
    if (world < 1 || rank < 0 || rank >= world || !ncol) return 1;
    const size_t nblk = (ncol + SFG_SLOTS - 1) / SFG_SLOTS, b0 = nblk * (size_t)rank / (size_t)world, b1 = nblk * ((size_t)rank + 1) / (size_t)world;
    if (blk0) *blk0 = b0; if (blk1) *blk1 = b1;
    if (col0) *col0 = b0 * SFG_SLOTS; if (col1) *col1 = std::min(b1 * (size_t)SFG_SLOTS, ncol);
    return 0;
}

// the engine's share of the configuration surface (ctx.hip has the rule): sfg_config, then the operator's environment; the forced exchange at world 1 is a test switch
static void mgpu_read_config(sfg_mgpu *mg, const sfg_config *pc) {
    if (pc && pc->struct_size >= offsetof(sfg_config, mgpu_transport) + sizeof(pc->mgpu_transport) && pc->mgpu_transport) mg->direct = !strcmp(pc->mgpu_transport, "direct");
    if (pc && pc->struct_size >= offsetof(sfg_config, mgpu_cache_bytes) + sizeof(pc->mgpu_cache_bytes) && pc->mgpu_cache_bytes) mg->cache_budget = pc->mgpu_cache_bytes == SIZE_MAX ? 0 : pc->mgpu_cache_bytes;
    if (pc && pc->struct_size >= offsetof(sfg_config, rccl_lib) + sizeof(pc->rccl_lib) && pc->rccl_lib) mg->rccl_lib = pc->rccl_lib;
    if (const char *e = getenv("SFG_MGPU_TRANSPORT")) mg->direct = !strcmp(e, "direct");
    if (const char *e = getenv("SFG_MGPU_CACHE_GB")) { double gb = atof(e); if (!(gb >= 0)) gb = 0; if (gb > 4096) gb = 4096; mg->cache_budget = (size_t)(gb * (double)(1ULL << 30)); }
    const char *hooks = getenv("SFG_ENABLE_TEST_HOOKS");
    if (hooks && atoi(hooks) == 1) if (const char *e = getenv("SFG_MGPU_FORCE_COLLECTIVES")) mg->force_coll = atoi(e) != 0;
}
static const char *rank_exec_init(MgRank &R) {
    if (hipSetDevice(R.device) != hipSuccess) return "hipSetDevice failed";
    // The collectives' queue is the context's encode queue (idle unless SFG_MM_ENC_OVERLAP=1 - then the engine makes its own).  Not a fourth queue by default: measured
    // (tools/r5_order.sh, tools/r5_collq.sh, profiles/r05_mgpu_queue_count.txt), with a fourth library stream IN USE and the product on the context's own queue every
    // kernel of the step starts 15 - 30 us later (a rank's step 1.65 s against 1.46 s); a stream that only exists costs nothing, and with GPU_MAX_HW_QUEUES <= 3, or with
    // the product on a stream the caller made, the effect vanishes - it depends on which hardware queues the runtime hands the streams.
#ifdef SFG_AB
    const char *cq = getenv("SFG_MGPU_COLL_QUEUE");       // diagnostics (tools/r5_collq.sh): "own" = a queue of the engine's whatever the schedule, "spare" = made but not used
    const bool own = cq && !strcmp(cq, "own"), spare = cq && !strcmp(cq, "spare");
#else
    const bool own = false, spare = false;
#endif
    const bool enc_busy = R.ctx->cfg.stage_pack && !R.ctx->cfg.stage_same_queue;      // (A/B build: the streamed transposition runs on the encode queue whatever the overlap switches say)
    if (!own && !enc_busy && (spare || R.ctx->cfg.no_enc_overlap || R.ctx->cfg.no_overlap)) {
        if (spare && hipStreamCreateWithFlags(&R.spare, hipStreamNonBlocking) != hipSuccess) return "hipStreamCreate failed";
        R.coll = R.ctx->enc_stream; R.own_coll = false;
    } else { if (hipStreamCreateWithFlags(&R.coll, hipStreamNonBlocking) != hipSuccess) return "hipStreamCreate failed"; R.own_coll = true; }
    for (int i = 0; i < 2; i++) if (hipEventCreateWithFlags(&R.ev_acc[i], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&R.ev_rs[i], hipEventDisableTiming) != hipSuccess) return "hipEventCreate failed";
    if (hipEventCreateWithFlags(&R.ev_c, hipEventDisableTiming) != hipSuccess) return "hipEventCreate failed";
    if (hipMalloc((void **)&R.status_dev, 64) != hipSuccess || hipHostMalloc((void **)&R.status_host, 64, hipHostMallocDefault) != hipSuccess) return "status word allocation failed";
    return nullptr;
}

extern "C" void sfg_mgpu_destroy(sfg_mgpu *mg) {
    if (!mg) return;
    for (auto &R : mg->r) {
        if (!R.ctx) continue;                          // (a rank whose context was never made - a bad device index - owns nothing, and its device must not be touched)
        (void)hipSetDevice(R.device);
        (void)sfg_sync_all(R.ctx);
        if (R.coll) (void)hipStreamSynchronize(R.coll);
    }
    for (auto &R : mg->r) if (R.comm) { (void)hipSetDevice(R.device); (void)g_rccl.CommDestroy(R.comm); }
    for (auto &R : mg->r) {
        if (!R.ctx) continue;
        (void)hipSetDevice(R.device);
        if (R.coll && R.own_coll) (void)hipStreamDestroy(R.coll);
        if (R.spare) (void)hipStreamDestroy(R.spare);
        for (int i = 0; i < 2; i++) { if (R.ev_acc[i]) (void)hipEventDestroy(R.ev_acc[i]); if (R.ev_rs[i]) (void)hipEventDestroy(R.ev_rs[i]); }
        if (R.ev_c) (void)hipEventDestroy(R.ev_c);
        if (R.status_dev) (void)hipFree(R.status_dev);
        if (R.status_host) (void)hipHostFree(R.status_host);
        if (R.ctx) sfg_ctx_destroy(R.ctx);
    }
    delete mg;
}

extern "C" int sfg_mgpu_unique_id(uint8_t *id128) {
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    { std::lock_guard<std::mutex> lk(g_rccl_mu); const std::string e = g_rccl.load(); if (!e.empty()) { g_mgpu_create_error = e; return 1; } }
    ncclUniqueId id; const ncclResult_t rc = g_rccl.GetUniqueId(&id);
    if (rc != ncclSuccess) { g_mgpu_create_error = std::string("ncclGetUniqueId: ") + g_rccl.GetErrorString(rc); return 1; }
    memcpy(id128, &id, 128); return 0;
}

static int mgpu_create_common(sfg_mgpu **out, const int *devices, int n, int rank0, int world, const uint8_t *id128, int logN, int nq, int np, const uint64_t *moduli,
                              const uint64_t *psi, double scale, const sfg_config *config) {
    *out = nullptr;
    if (n < 1 || n > 64 || world < n || world > 64 * 1024) { g_mgpu_create_error = "sfg_mgpu_create: bad device / rank counts"; return 1; }
    sfg_mgpu *mg = new sfg_mgpu();
    mg->world = world; mg->single_process = id128 == nullptr; mg->rv.n = n;
    mgpu_read_config(mg, config);
    auto fail = [&](const std::string &m) { g_mgpu_create_error = m; sfg_mgpu_destroy(mg); return 1; };
#ifdef SFG_AB
    if (const char *e = getenv("SFG_MGPU_SOLO")) {
        int r = 0, w = 0;
        if (sscanf(e, "%d/%d", &r, &w) != 2 || w < 1 || r < 0 || r >= w || n != 1 || id128) return fail("SFG_MGPU_SOLO=r/w needs one local device and a single process");
        mg->solo = true; mg->world = world = w; rank0 = r;
    }
#endif
    bool dup = false;
    for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) dup = dup || devices[i] == devices[j];
    if (dup) mg->direct = true;                      // several ranks on one device (RCCL refuses that): the in-process transport
    if (mg->direct && !mg->single_process) return fail("sfg_mgpu_create_rank: the direct transport exists inside one process only");
    mg->r.resize((size_t)n);
    for (int i = 0; i < n; i++) {
        MgRank &R = mg->r[(size_t)i]; R.rank = rank0 + i; R.device = devices[i];
        if (sfg_ctx_create_ex(&R.ctx, devices[i], logN, nq, np, moduli, psi, scale, config)) return fail(std::string("sfg_mgpu_create: device ") + std::to_string(devices[i]) + ": " + sfg_last_error(nullptr));
        if (const char *e = rank_exec_init(R)) return fail(e);
    }
    const bool need_comm = (world > 1 || mg->force_coll) && !mg->solo;
    if (need_comm && !mg->direct) {
        { std::lock_guard<std::mutex> lk(g_rccl_mu); const std::string e = g_rccl.load(mg->rccl_lib.empty() ? nullptr : mg->rccl_lib.c_str()); if (!e.empty()) return fail(e); }
        if (mg->single_process) {
            std::vector<ncclComm_t> comms((size_t)n);
            const ncclResult_t rc = g_rccl.CommInitAll(comms.data(), n, devices);
            if (rc != ncclSuccess) return fail(std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(rc));
            for (int i = 0; i < n; i++) mg->r[(size_t)i].comm = comms[(size_t)i];
        } else {
            ncclUniqueId id; memcpy(&id, id128, 128);
            if (hipSetDevice(devices[0]) != hipSuccess) return fail("hipSetDevice failed");
            const ncclResult_t rc = g_rccl.CommInitRank(&mg->r[0].comm, world, id, rank0);
            if (rc != ncclSuccess) return fail(std::string("ncclCommInitRank: ") + g_rccl.GetErrorString(rc));
        }
    }
    if (mg->direct) for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) if (devices[i] != devices[j]) {
        int can = 0; (void)hipDeviceCanAccessPeer(&can, devices[i], devices[j]);
        if (!can) return fail("sfg_mgpu_create: direct transport: devices " + std::to_string(devices[i]) + " and " + std::to_string(devices[j]) + " cannot access each other's memory");
        (void)hipSetDevice(devices[i]);
        const hipError_t e = hipDeviceEnablePeerAccess(devices[j], 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) return fail(std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
        (void)hipGetLastError();
    }
    *out = mg;
    return 0;
}
extern "C" int sfg_mgpu_create(sfg_mgpu **out, const int *devices, int n, int logN, int nq, int np, const uint64_t *moduli, const uint64_t *psi, double scale) {
    if (!devices) { g_mgpu_create_error = "sfg_mgpu_create: null device list"; *out = nullptr; return 1; }
    return mgpu_create_common(out, devices, n, 0, n, nullptr, logN, nq, np, moduli, psi, scale, nullptr);
}
extern "C" int sfg_mgpu_create_ex(sfg_mgpu **out, const int *devices, int n, int logN, int nq, int np, const uint64_t *moduli, const uint64_t *psi, double scale,
                                  const sfg_config *config) {
    if (!out) { g_mgpu_create_error = "sfg_mgpu_create: null result pointer"; return 1; }
    if (!devices) { g_mgpu_create_error = "sfg_mgpu_create: null device list"; *out = nullptr; return 1; }
    return mgpu_create_common(out, devices, n, 0, n, nullptr, logN, nq, np, moduli, psi, scale, config);
}
extern "C" int sfg_mgpu_create_rank(sfg_mgpu **out, int device, int rank, int world, const uint8_t *id128, int logN, int nq, int np, const uint64_t *moduli,
                                    const uint64_t *psi, double scale) {
    if (!id128 || rank < 0 || rank >= world) { g_mgpu_create_error = "sfg_mgpu_create_rank: bad rank / missing id"; *out = nullptr; return 1; }
    return mgpu_create_common(out, &device, 1, rank, world, id128, logN, nq, np, moduli, psi, scale, nullptr);
}
extern "C" int sfg_mgpu_create_rank_ex(sfg_mgpu **out, int device, int rank, int world, const uint8_t *id128, int logN, int nq, int np, const uint64_t *moduli,
                                       const uint64_t *psi, double scale, const sfg_config *config) {
    if (!out) { g_mgpu_create_error = "sfg_mgpu_create_rank: null result pointer"; return 1; }
    if (!id128 || rank < 0 || rank >= world) { g_mgpu_create_error = "sfg_mgpu_create_rank: bad rank / missing id"; *out = nullptr; return 1; }
    return mgpu_create_common(out, &device, 1, rank, world, id128, logN, nq, np, moduli, psi, scale, config);
}
extern "C" const char *sfg_mgpu_last_error(const sfg_mgpu *mg) { return mg ? mg->err.c_str() : g_mgpu_create_error.c_str(); }
extern "C" int sfg_mgpu_world(const sfg_mgpu *mg) { return mg ? mg->world : 0; }
extern "C" int sfg_mgpu_nlocal(const sfg_mgpu *mg) { return mg ? (int)mg->r.size() : 0; }
extern "C" int sfg_mgpu_rank(const sfg_mgpu *mg, int local) { return mg && local >= 0 && local < (int)mg->r.size() ? mg->r[(size_t)local].rank : -1; }
extern "C" sfg_ctx *sfg_mgpu_ctx(sfg_mgpu *mg, int local) { return mg && local >= 0 && local < (int)mg->r.size() ? mg->r[(size_t)local].ctx : nullptr; }
extern "C" const char *sfg_mgpu_transport(const sfg_mgpu *mg) { return !mg ? "" : mg->solo ? "solo" : mg->world == 1 && !mg->force_coll ? "none" : mg->direct ? "direct" : "rccl"; }
// What the communicator itself reports for local rank `local` (ncclCommCount / ncclCommUserRank): a record of a multi-GPU run can then state "RCCL saw N ranks"
// instead of inferring it from the launcher's environment.  *nranks = *rank = 0 when the engine holds no communicator (world 1, direct transport, solo timing).
extern "C" int sfg_mgpu_comm_info(sfg_mgpu *mg, int local, int *nranks, int *rank) {
    MG_NEED(mg, mg != nullptr, "null engine");
    MG_NEED(mg, local >= 0 && local < (int)mg->r.size(), "local rank out of range");
    if (nranks) *nranks = 0; if (rank) *rank = 0;
    MgRank &R = mg->r[(size_t)local];
    if (!R.comm) return 0;
    if (!g_rccl.CommCount || !g_rccl.CommUserRank) MG_FAIL(mg, "sfg_mgpu_comm_info: this RCCL build exports no ncclCommCount / ncclCommUserRank");
    int n = 0, r = 0;
    if (g_rccl.CommCount(R.comm, &n) != ncclSuccess || g_rccl.CommUserRank(R.comm, &r) != ncclSuccess) MG_FAIL(mg, "sfg_mgpu_comm_info: ncclCommCount / ncclCommUserRank failed");
    if (nranks) *nranks = n; if (rank) *rank = r;
    return 0;
}

extern "C" int sfg_mgpu_load_rotkey(sfg_mgpu *mg, uint64_t galois_el, const uint64_t *key_host, int montgomery_form) {
    MG_NEED(mg, mg != nullptr, "null engine");
    return run_ranks(mg, [&](MgRank &R, int) { R_CTX(R, sfg_ctx_load_rotkey(R.ctx, galois_el, key_host, montgomery_form)); return 0; });
}
extern "C" int sfg_mgpu_load_relinkey(sfg_mgpu *mg, const uint64_t *key_host, int montgomery_form) {
    MG_NEED(mg, mg != nullptr, "null engine");
    return run_ranks(mg, [&](MgRank &R, int) { R_CTX(R, sfg_ctx_load_relinkey(R.ctx, key_host, montgomery_form)); return 0; });
}
extern "C" int sfg_mgpu_fill_rotkeys_synthetic(sfg_mgpu *mg, const int *rot_left, int nrot, uint64_t seed) {
    MG_NEED(mg, mg != nullptr, "null engine");
    return run_ranks(mg, [&](MgRank &R, int) { R_CTX(R, sfg_fill_rotkeys_synthetic(R.ctx, rot_left, nrot, seed)); return 0; });
}
extern "C" int sfg_mgpu_synchronize(sfg_mgpu *mg) {
    MG_NEED(mg, mg != nullptr, "null engine");
    return run_ranks(mg, [&](MgRank &R, int) { R_HIP(R, hipSetDevice(R.device)); R_HIP(R, hipStreamSynchronize(R.coll)); R_CTX(R, sfg_ctx_synchronize(R.ctx)); return 0; });
}

// ---------------------------------------------------------------- the sharded genotype matrix
static sfg_mgeno *mgeno_new(sfg_mgpu *mg, size_t nrow, size_t ncol) {
    sfg_mgeno *g = new sfg_mgeno(); g->nrow = nrow; g->ncol = ncol;
    const size_t n = mg->r.size(); g->shard.assign(n, nullptr); g->owned.assign(n, nullptr); g->blk0.assign(n, 0); g->blk1.assign(n, 0);
    return g;
}
extern "C" void sfg_mgpu_geno_free(sfg_mgpu *mg, sfg_mgeno *g) {
    if (!g) return;
    if (!mg) { delete g; return; }            // (nothing to free the shards with: the engine that made them is gone and took its contexts' memory along)
    for (size_t i = 0; i < g->shard.size() && i < mg->r.size(); i++) {
        if (g->shard[i]) sfg_geno_free(mg->r[i].ctx, g->shard[i]);
        if (g->owned[i]) (void)sfg_free(mg->r[i].ctx, g->owned[i]);
    }
    delete g;
}
// geno_host: the party's WHOLE matrix, row-major int8 with row stride ld (what GenoFileStream delivers); every local rank uploads its own column window
extern "C" int sfg_mgpu_geno_upload(sfg_mgpu *mg, const int8_t *geno_host, size_t nrow, size_t ncol, size_t ld, sfg_mgeno **out) {
    MG_NEED(mg, mg != nullptr && out != nullptr, "null engine / result pointer");
    *out = nullptr;
    if (!geno_host || !nrow || !ncol || ld < ncol) MG_FAIL(mg, "sfg_mgpu_geno_upload: bad dimensions");
    sfg_mgeno *g = mgeno_new(mg, nrow, ncol);
    const int rc = run_ranks(mg, [&](MgRank &R, int i) {
        size_t c0, c1; (void)sfg_mgpu_shard(mg->world, ncol, R.rank, &g->blk0[(size_t)i], &g->blk1[(size_t)i], &c0, &c1);
        if (c1 > c0) R_CTX(R, sfg_geno_upload(R.ctx, geno_host + c0, nrow, c1 - c0, ld, &g->shard[(size_t)i]));
        return 0;
    });
    if (rc) { sfg_mgpu_geno_free(mg, g); return 1; }
    *out = g; return 0;
}
// Row-streamed form (MatMult4StreamPreprocess reads one row at a time: matmult.go:914-1041, filestream.go:414-426): _create makes every rank's window, _write_rows
// scatters a chunk of whole-matrix rows to the ranks' column windows, _compare_rows compares a chunk of the rows of the matrix (or, with SFG_TRANSPOSE, of its
// transpose: pca.go:113 registers X^T from a second file) with the resident shards on the devices and adds the number of differing entries to *ndiff.
extern "C" int sfg_mgpu_geno_create(sfg_mgpu *mg, size_t nrow, size_t ncol, sfg_mgeno **out) {
    MG_NEED(mg, mg != nullptr && out != nullptr, "null engine / result pointer");
    *out = nullptr;
    if (!nrow || !ncol) MG_FAIL(mg, "sfg_mgpu_geno_create: bad dimensions");
    sfg_mgeno *g = mgeno_new(mg, nrow, ncol);
    const int rc = run_ranks(mg, [&](MgRank &R, int i) {
        size_t c0, c1; (void)sfg_mgpu_shard(mg->world, ncol, R.rank, &g->blk0[(size_t)i], &g->blk1[(size_t)i], &c0, &c1);
        if (c1 > c0) R_CTX(R, sfg_geno_create(R.ctx, nrow, c1 - c0, &g->shard[(size_t)i]));
        return 0;
    });
    if (rc) { sfg_mgpu_geno_free(mg, g); return 1; }
    *out = g; return 0;
}
extern "C" int sfg_mgpu_geno_write_rows(sfg_mgpu *mg, sfg_mgeno *g, size_t row0, size_t nrows, const int8_t *rows_host, size_t ld) {
    MG_NEED(mg, mg != nullptr, "null engine");
    MG_NEED(mg, g && g->shard.size() == mg->r.size(), "the matrix belongs to another engine");
    if (!nrows) return 0;
    if (!rows_host || ld < g->ncol || row0 > g->nrow || nrows > g->nrow - row0) MG_FAIL(mg, "sfg_mgpu_geno_write_rows: rows [%zu, %zu) of a %zu x %zu matrix, row stride %zu", row0, row0 + nrows, g->nrow, g->ncol, ld);
    return run_ranks(mg, [&](MgRank &R, int i) {
        if (g->shard[(size_t)i]) R_CTX(R, sfg_geno_write_rows(R.ctx, g->shard[(size_t)i], row0, nrows, rows_host + g->blk0[(size_t)i] * SFG_SLOTS, ld));
        return 0;
    });
}
extern "C" int sfg_mgpu_geno_compare_rows(sfg_mgpu *mg, const sfg_mgeno *g, unsigned flags, size_t row0, size_t nrows, const int8_t *rows_host, size_t ld, uint64_t *ndiff) {
    MG_NEED(mg, mg != nullptr, "null engine");
    MG_NEED(mg, g && g->shard.size() == mg->r.size() && ndiff, "the matrix belongs to another engine / null result pointer");
    const bool tr = flags & SFG_TRANSPOSE;
    const size_t nrow_l = tr ? g->ncol : g->nrow, ncol_l = tr ? g->nrow : g->ncol;
    if (!nrows) return 0;
    if (!rows_host || ld < ncol_l || row0 > nrow_l || nrows > nrow_l - row0) MG_FAIL(mg, "sfg_mgpu_geno_compare_rows: rows [%zu, %zu) of a %zu x %zu matrix, row stride %zu", row0, row0 + nrows, nrow_l, ncol_l, ld);
    std::vector<uint64_t> bad(mg->r.size(), 0);
    const int rc = run_ranks(mg, [&](MgRank &R, int i) {
        const sfg_geno *sh = g->shard[(size_t)i];
        if (!sh) return 0;
        const size_t c0 = g->blk0[(size_t)i] * SFG_SLOTS, c1 = c0 + sh->ncol;          // the rank's window of stored columns
        if (!tr) { R_CTX(R, sfg_geno_compare_rows(R.ctx, sh, 0, row0, nrows, rows_host + c0, ld, &bad[(size_t)i])); return 0; }
        // rows of the transpose are stored COLUMNS: this rank answers for the rows that fall into its window
        const size_t lo = std::max(row0, c0), hi = std::min(row0 + nrows, c1);
        if (lo < hi) R_CTX(R, sfg_geno_compare_rows(R.ctx, sh, SFG_TRANSPOSE, lo - c0, hi - lo, rows_host + (lo - row0) * ld, ld, &bad[(size_t)i]));
        return 0;
    });
    if (rc) return rc;
    for (uint64_t b : bad) *ndiff += b;
    return 0;
}
// per-rank handles the caller made on the ranks' own contexts (sfg_geno_from_bed / _from_pgen / _from_device of the rank's window); ownership passes to the result.
// shards[i] == NULL for a local rank whose window is empty.
extern "C" int sfg_mgpu_geno_adopt(sfg_mgpu *mg, size_t nrow, size_t ncol, sfg_geno *const *shards, sfg_mgeno **out) {
    MG_NEED(mg, mg != nullptr && out != nullptr, "null engine / result pointer");
    *out = nullptr;
    MG_NEED(mg, shards != nullptr && nrow && ncol, "null shard list / empty matrix");
    sfg_mgeno *g = mgeno_new(mg, nrow, ncol);
    for (size_t i = 0; i < mg->r.size(); i++) {
        size_t c0, c1; (void)sfg_mgpu_shard(mg->world, ncol, mg->r[i].rank, &g->blk0[i], &g->blk1[i], &c0, &c1);
        if ((c1 > c0) != (shards[i] != nullptr) || (shards[i] && (shards[i]->nrow != nrow || shards[i]->ncol != c1 - c0))) {
            delete g; MG_FAIL(mg, "sfg_mgpu_geno_adopt: shard %zu is not the %zu x %zu window [%zu, %zu) of rank %d", i, nrow, c1 - c0, c0, c1, mg->r[i].rank); }
    }
    for (size_t i = 0; i < mg->r.size(); i++) g->shard[i] = shards[i];
    *out = g; return 0;
}
// bench.py / tests: every rank generates exactly the window of the SAME global synthetic matrix it owns (sfg_fill_geno_window_dev), so any world size multiplies
// the same matrix.  packed != 0: 2-bit residency (sfg_geno_pack)
extern "C" int sfg_mgpu_geno_synthetic(sfg_mgpu *mg, size_t nrow, size_t ncol, uint64_t seed, int packed, sfg_mgeno **out) {
    MG_NEED(mg, mg != nullptr && out != nullptr, "null engine / result pointer");
    *out = nullptr;
    if (!nrow || !ncol) MG_FAIL(mg, "sfg_mgpu_geno_synthetic: bad dimensions");
    sfg_mgeno *g = mgeno_new(mg, nrow, ncol);
    const int rc = run_ranks(mg, [&](MgRank &R, int i) {
        size_t c0, c1; (void)sfg_mgpu_shard(mg->world, ncol, R.rank, &g->blk0[(size_t)i], &g->blk1[(size_t)i], &c0, &c1);
        if (c1 <= c0) return 0;
        const size_t w = c1 - c0; void *buf = nullptr;
        R_CTX(R, sfg_malloc(R.ctx, &buf, nrow * w));
        g->owned[(size_t)i] = buf;
        R_CTX(R, sfg_fill_geno_window_dev(R.ctx, (int8_t *)buf, nrow, w, w, c0, ncol, seed));
        R_CTX(R, sfg_geno_from_device(R.ctx, (const int8_t *)buf, nrow, w, w, &g->shard[(size_t)i]));
        if (packed) {
            sfg_geno *p = nullptr;
            R_CTX(R, sfg_geno_pack(R.ctx, g->shard[(size_t)i], &p));
            sfg_geno_free(R.ctx, g->shard[(size_t)i]); g->shard[(size_t)i] = p;
            R_CTX(R, sfg_free(R.ctx, buf)); g->owned[(size_t)i] = nullptr;
        }
        return 0;
    });
    if (rc) { sfg_mgpu_geno_free(mg, g); return 1; }
    *out = g; return 0;
}
extern "C" const sfg_geno *sfg_mgpu_geno_shard(const sfg_mgeno *g, int local) { return g && local >= 0 && (size_t)local < g->shard.size() ? g->shard[(size_t)local] : nullptr; }
extern "C" int sfg_mgpu_geno_dims(const sfg_mgeno *g, size_t *nrow, size_t *ncol) { if (!g) return 1; if (nrow) *nrow = g->nrow; if (ncol) *ncol = g->ncol; return 0; }
extern "C" int sfg_mgpu_geno_blocks(const sfg_mgeno *g, int local, size_t *blk0, size_t *blk1) {
    if (!g || local < 0 || (size_t)local >= g->shard.size()) return 1;
    if (blk0) *blk0 = g->blk0[(size_t)local]; if (blk1) *blk1 = g->blk1[(size_t)local]; return 0;
}
extern "C" int sfg_mgpu_geno_set_plaintext_cache(sfg_mgpu *mg, const sfg_mgeno *g, size_t max_bytes_per_rank) {
    MG_NEED(mg, mg != nullptr, "null engine");
    MG_NEED(mg, g && g->shard.size() == mg->r.size(), "the matrix belongs to another engine");
    return run_ranks(mg, [&](MgRank &R, int i) { if (g->shard[(size_t)i]) R_CTX(R, sfg_geno_set_plaintext_cache(R.ctx, g->shard[(size_t)i], max_bytes_per_rank)); return 0; });
}

// ---------------------------------------------------------------- collectives (enqueued on `st` of the calling rank, in order with it)
static int coll_reduce_scatter(sfg_mgpu *mg, MgRank &R, const uint64_t *send, uint64_t *recv, size_t recv_count, hipStream_t st) {
    if (mg->solo) { R_HIP(R, hipMemcpyAsync(recv, send + (size_t)R.rank * recv_count, recv_count * 8, hipMemcpyDeviceToDevice, st)); return 0; }
    if (!mg->direct) { R_NCCL(R, g_rccl.ReduceScatter(send, recv, recv_count, ncclUint64, ncclSum, R.comm, st)); return 0; }
    const int n = mg->world;
    R_HIP(R, hipStreamSynchronize(st));                               // this rank's contribution is complete
    mg->rv.ptr[R.rank] = send;
    if (!mg->rv.barrier()) R_FAIL(R, "a peer rank failed");
    PeerPtrs pp; for (int p = 0; p < n; p++) pp.p[p] = (const u64 *)mg->rv.ptr[p] + (size_t)R.rank * recv_count;
    hipLaunchKernelGGL(k_sum_peers, dim3((unsigned)std::min<size_t>((recv_count + 255) / 256, 4096)), dim3(256), 0, st, (u64 *)recv, pp, n, recv_count);
    R_HIP(R, hipGetLastError());
    R_HIP(R, hipStreamSynchronize(st));
    if (!mg->rv.barrier()) R_FAIL(R, "a peer rank failed");          // everybody has read: the send buffers may be overwritten
    return 0;
}
static int coll_all_reduce(sfg_mgpu *mg, MgRank &R, uint64_t *buf, size_t count, hipStream_t st) {
    if (mg->solo) return 0;
    if (!mg->direct) { R_NCCL(R, g_rccl.AllReduce(buf, buf, count, ncclUint64, ncclSum, R.comm, st)); return 0; }
    const int n = mg->world;
    // in place: sum into a private copy first (a peer may still be reading this rank's buffer), swap after the second meeting
    uint64_t *tmp = nullptr;
    R_CTX(R, sfg_scratch(R.ctx, "mg.ar", count * 8, (void **)&tmp));
    R_HIP(R, hipStreamSynchronize(st));
    mg->rv.ptr[R.rank] = buf;
    if (!mg->rv.barrier()) R_FAIL(R, "a peer rank failed");
    PeerPtrs pp; for (int p = 0; p < n; p++) pp.p[p] = (const u64 *)mg->rv.ptr[p];
    hipLaunchKernelGGL(k_sum_peers, dim3((unsigned)std::min<size_t>((count + 255) / 256, 4096)), dim3(256), 0, st, (u64 *)tmp, pp, n, count);
    R_HIP(R, hipGetLastError());
    R_HIP(R, hipStreamSynchronize(st));
    if (!mg->rv.barrier()) R_FAIL(R, "a peer rank failed");
    R_HIP(R, hipMemcpyAsync(buf, tmp, count * 8, hipMemcpyDeviceToDevice, st));
    return 0;
}

// ---------------------------------------------------------------- agreement before a call's first exchange
// A rank that fails BEFORE it has enqueued its part of a collective (out of memory in a scratch pool, a context error) would leave its peers waiting in theirs for
// ever: RCCL has no timeout, and run_ranks joins every thread.  So every exchanging call has an agreement point after its allocations and rotation caches and before
// its first collective: each rank brings its status so far, all leave with the same verdict.  In-process ranks meet at the host rendezvous; one-rank-per-process
// worlds all-reduce a status word through the communicator itself (a rank that failed still takes part - that is the point) and read it back.  After the agreement
// point a failing rank ABORTS its communicator (ncclCommAbort) so that the peers' pending collectives return with an error instead of hanging, and the engine
// refuses further exchanges (`broken`).
static int coll_agree(sfg_mgpu *mg, MgRank &R, int local_rc) {
    if (mg->solo || (mg->world == 1 && !mg->force_coll)) return local_rc;
    if (mg->r.size() > 1) {
        if (local_rc) mg->rv.fail();
        if (!mg->rv.barrier() && !local_rc) { R.err = "a peer rank failed"; local_rc = 1; }
    }
    if (mg->single_process || mg->direct || !R.comm) return local_rc;
    if (hipSetDevice(R.device) != hipSuccess) return 1;
    *R.status_host = local_rc ? 1u : 0u;
    hipStream_t st = R.coll;
    if (hipMemcpyAsync(R.status_dev, R.status_host, 8, hipMemcpyHostToDevice, st) != hipSuccess) { if (!local_rc) R.err = "agreement: status upload failed"; return 1; }
    const ncclResult_t e = g_rccl.AllReduce(R.status_dev, R.status_dev, 1, ncclUint64, ncclSum, R.comm, st);
    if (e != ncclSuccess) { if (!local_rc) R.err = std::string("agreement all-reduce failed: ") + g_rccl.GetErrorString(e); return 1; }
    if (hipMemcpyAsync(R.status_host, R.status_dev, 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) { if (!local_rc) R.err = "agreement: status read-back failed"; return 1; }
    if (*R.status_host && !local_rc) { R.err = "a peer rank failed"; return 1; }
    return local_rc;
}
static void coll_abort(sfg_mgpu *mg, MgRank &R) {
    mg->broken = true;
    if (mg->direct) { mg->rv.fail(); return; }
    if (R.comm && g_rccl.CommAbort) { (void)hipSetDevice(R.device); (void)g_rccl.CommAbort(R.comm); R.comm = nullptr; }
}

// Pre-flight of the exchange paths (bench.py --gpus N runs it before the warm-up; a Go party may after hip.Init): a reduce-scatter and an all-reduce of a known
// uint64 pattern through the very functions the products use, on the collectives' queue of every rank, checked on the host.  rank r contributes
// v[x] = (r + 1) * 2^40 + x; the sum over ranks at position x is world (world + 1) / 2 * 2^40 + world x.  count_per_rank words per rank slice (>= 1).
extern "C" int sfg_mgpu_preflight(sfg_mgpu *mg, size_t count_per_rank) {
    MG_NEED(mg, mg != nullptr, "null engine");
    if (mg->broken) MG_FAIL(mg, "sfg_mgpu_preflight: the engine's communicator was aborted by an earlier failure");
    if (!count_per_rank || count_per_rank > (1u << 24)) MG_FAIL(mg, "sfg_mgpu_preflight: count_per_rank out of range");
    if (mg->solo || (mg->world == 1 && !mg->force_coll)) return 0;
    MgApiScope scope(mg);
    const size_t w = (size_t)mg->world, total = w * count_per_rank;
    return run_ranks(mg, [&](MgRank &R, int) {
        R_HIP(R, hipSetDevice(R.device));
        uint64_t *send = nullptr, *recv = nullptr; int rc = 0;
        std::vector<uint64_t> h(total);
        for (size_t x = 0; x < total; x++) h[x] = ((uint64_t)(R.rank + 1) << 40) + x;
        if (sfg_scratch(R.ctx, "mg.pf_send", total * 8, (void **)&send) || sfg_scratch(R.ctx, "mg.pf_recv", total * 8, (void **)&recv)) { R.err = R.ctx->err; rc = 1; }
        if (!rc && hipMemcpy(send, h.data(), total * 8, hipMemcpyHostToDevice) != hipSuccess) { R.err = "preflight upload failed"; rc = 1; }
        if (coll_agree(mg, R, rc)) return 1;
        const uint64_t base = (uint64_t)(w * (w + 1) / 2) << 40;
        auto after = [&](int e) { if (e) coll_abort(mg, R); return e; };
        if (after(coll_reduce_scatter(mg, R, send, recv, count_per_rank, R.coll))) return 1;
        if (hipMemcpyAsync(h.data(), recv, count_per_rank * 8, hipMemcpyDeviceToHost, R.coll) != hipSuccess || hipStreamSynchronize(R.coll) != hipSuccess) { R.err = "preflight read-back failed"; return after(1); }
        for (size_t x = 0; x < count_per_rank; x++) if (h[x] != base + w * ((size_t)R.rank * count_per_rank + x)) {
            R.err = "preflight: reduce-scatter returned a wrong word at " + std::to_string(x); return after(1); }
        if (after(coll_all_reduce(mg, R, send, total, R.coll))) return 1;
        if (hipMemcpyAsync(h.data(), send, total * 8, hipMemcpyDeviceToHost, R.coll) != hipSuccess || hipStreamSynchronize(R.coll) != hipSuccess) { R.err = "preflight read-back failed"; return after(1); }
        for (size_t x = 0; x < total; x++) if (h[x] != base + w * x) { R.err = "preflight: all-reduce returned a wrong word at " + std::to_string(x); return after(1); }
        return 0;
    });
}

// ---------------------------------------------------------------- the products
// Q' * X^T of one rank: see the header of this file
static int rank_contract(sfg_mgpu *mg, MgRank &R, int li, const uint64_t *A, int s, int in_level, int L, const sfg_mgeno *g, unsigned flags, uint64_t *out) {
    sfg_ctx *ctx = R.ctx;
    R_HIP(R, hipSetDevice(R.device));
    ApiScope api_scope(ctx);                             // one top-level call for the scratch pools' bookkeeping
    const int world = mg->world, d = SFG_D, N = SFG_N;
    const sfg_geno *shard = g->shard[(size_t)li];
    const int nloc = (int)(g->blk1[(size_t)li] - g->blk0[(size_t)li]), nbr_x = (int)((g->nrow + SFG_SLOTS - 1) / SFG_SLOTS);
    const unsigned fl = (flags & SFG_SQUARE) | SFG_TRANSPOSE;
    const size_t outw = (size_t)2 * L * N, accw = (size_t)s * outw;
    const int gpr = (d + world - 1) / world, g_lo = R.rank * gpr;
    const size_t col = (size_t)d * accw, colp = (size_t)world * gpr * accw, mine = (size_t)gpr * accw;
    if (world == 1 && !mg->force_coll) { R_CTX(R, sfg_matmul_resident_dev(ctx, A, s, in_level, L, shard, fl, out)); return 0; }
    I8RotPre pre8;
    bool pipe = false;
    int PW = 1;                                          // block columns per multiply call of the pipeline: with int8 rot tiles TWO, so that the second column's encode carries the
                                                         // first one's plaintext transposition (kernels.hpp PtRide; a one-column call of one MAC group has nothing to ride in)
    uint64_t *acc_mine = nullptr, *acc2 = nullptr; double *cache = nullptr;
    hipStream_t cs = ctx->stream;
    size_t acc_w = 0;
    // ---- everything that allocates or can fail on this rank's own account, BEFORE the first exchange
    auto prepare = [&]() -> int {
        if (mg->broken) R_FAIL(R, "sfg_mgpu_matmul: the engine's communicator was aborted by an earlier failure");
        if (in_level < L) R_FAIL(R, "sfg_mgpu_matmul: input level %d below max_level %d", in_level, L);
        size_t jobw = 0, tailw = 0;
        R_CTX(R, sfg_rotcache_layout(ctx, s, L, &jobw, &tailw));
        const size_t cache_w = (size_t)nloc * s * jobw + tailw;
        // The rank's own baby-step rotations, once per product, for the per-column pipeline: as the int8 MAC's rot TILES where the context multiplies on the matrix core
        // (1.3 GB per block row at s = 15; every column then multiplies there whatever the number of MAC groups - with fp64 rows a rank of more than two groups, i.e. a world
        // of 4 or fewer at 100k x 1M, fell back to the fp64 kernel: 1.04 s of a 3.28 s rank step), else as fp64 operand rows (2.15 GB per block row) while they fit
        if (nloc) R_CTX(R, i8_rotpre_build(ctx, (const u64 *)A, s, in_level, L, nloc, nullptr, mg->cache_budget, "mg.rot8", pre8));
        pipe = !nloc || pre8.G || cache_w * 8 <= mg->cache_budget;
        R_CTX(R, sfg_scratch(ctx, "mg.mine", (size_t)nbr_x * mine * 8, (void **)&acc_mine));
        if (mg->direct) { uint64_t *tmp = nullptr; R_CTX(R, sfg_scratch(ctx, "mg.ar", (size_t)s * nbr_x * outw * 8, (void **)&tmp)); }      // (the direct all-reduce's private copy: not after the agreement)
        // (the collectives' queue must not start before earlier work of the compute queue that still reads these buffers: previous call's finalize)
        R_HIP(R, hipEventRecord(R.ev_c, cs)); R_HIP(R, hipStreamWaitEvent(R.coll, R.ev_c, 0));
        if (pipe) {
            PW = pre8.G ? 2 : 1;
            const size_t acc2_bytes = (size_t)2 * PW * colp * 8;
            const bool fresh = ctx->pool.find("mg.acc2") == ctx->pool.end() || ctx->pool["mg.acc2"].second < acc2_bytes;
            R_CTX(R, sfg_scratch(ctx, "mg.acc2", acc2_bytes, (void **)&acc2));
            if (fresh || !nloc) R_HIP(R, hipMemsetAsync(acc2, 0, acc2_bytes, cs));       // the padded giant slots (>= 91) are never written by a product: zero once
            if (nloc && !pre8.G) {
                R_CTX(R, sfg_scratch(ctx, "mg.cache", cache_w * 8, (void **)&cache));
                R_CTX(R, sfg_rotcache_build_rows_dev(ctx, A, s, in_level, L, nloc, 0, nloc, cache));
            }
            // the first column(s) are multiplied before the agreement too: that grows the product's own scratch pools (panel, tiles, accumulators) to their final shape
            if (nloc && pre8.G) R_CTX(R, matmul_accumulate_i8pre(ctx, pre8, s, L, shard, fl, 0, std::min(PW, nbr_x), 0, acc2, colp));
            else if (nloc) R_CTX(R, sfg_matmul_accumulate_rc_dev(ctx, cache, s, L, shard, fl, 0, nloc, 0, 1, 0, acc2));
        } else {                                           // the rank's own cache would not fit: the library's grouped rotation cache, reduce-scatters after the product
            acc_w = ((size_t)nbr_x * d + ((size_t)world * gpr - d)) * accw;
            R_CTX(R, sfg_scratch(ctx, "mg.acc2", acc_w * 8, (void **)&acc2));
            R_HIP(R, hipMemsetAsync(acc2, 0, acc_w * 8, cs));
            if (nloc) R_CTX(R, sfg_matmul_accumulate_dev(ctx, A, s, in_level, L, shard, fl, 0, nloc, 0, nbr_x, 0, acc2));
        }
        return 0;
    };
    if (coll_agree(mg, R, prepare())) { i8_rotpre_free(pre8); return 1; }
    // ---- from here on a failure aborts the communicator: the peers are inside (or about to enter) their collectives
    auto exchange = [&]() -> int {
        if (pipe) {
            for (int j = 0, p = 0; j < nbr_x; j += PW, p++) {      // columns [j, j + PW) are multiplied while the previous PW columns are reduce-scattered
                const int je = std::min(nbr_x, j + PW);
                uint64_t *buf = acc2 + (size_t)(p & 1) * PW * colp;
                if (p >= 2) R_HIP(R, hipStreamWaitEvent(cs, R.ev_rs[p & 1], 0));            // the reduce-scatters of the call before last have read this buffer
                if (j > 0) {
                    if (nloc && pre8.G) R_CTX(R, matmul_accumulate_i8pre(ctx, pre8, s, L, shard, fl, j, je, 0, buf, colp));
                    else if (nloc) R_CTX(R, sfg_matmul_accumulate_rc_dev(ctx, cache, s, L, shard, fl, 0, nloc, j, je, 0, buf));
                }
                R_HIP(R, hipEventRecord(R.ev_acc[p & 1], cs)); R_HIP(R, hipStreamWaitEvent(R.coll, R.ev_acc[p & 1], 0));
                for (int c = j; c < je; c++)
                    if (coll_reduce_scatter(mg, R, buf + (size_t)(c - j) * colp, acc_mine + (size_t)c * mine, mine, R.coll)) return 1;
                R_HIP(R, hipEventRecord(R.ev_rs[p & 1], R.coll));
            }
        } else {
            R_HIP(R, hipEventRecord(R.ev_acc[0], cs)); R_HIP(R, hipStreamWaitEvent(R.coll, R.ev_acc[0], 0));
            for (int j = 0; j < nbr_x; j++)                 // the window of the last giants runs into the next block column: those slots are ignored by the finalize
                if (coll_reduce_scatter(mg, R, acc2 + (size_t)j * col, acc_mine + (size_t)j * mine, mine, R.coll)) return 1;
        }
        R_HIP(R, hipEventRecord(R.ev_c, R.coll)); R_HIP(R, hipStreamWaitEvent(cs, R.ev_c, 0));
        R_CTX(R, sfg_reduce_rows_dev(ctx, acc_mine, (size_t)nbr_x * gpr * s * 2, L));
        R_CTX(R, sfg_matmul_finalize_slots_dev(ctx, acc_mine, s, L, nbr_x, gpr, g_lo, 0, gpr, 0, out));
        R_HIP(R, hipEventRecord(R.ev_c, cs)); R_HIP(R, hipStreamWaitEvent(R.coll, R.ev_c, 0));
        if (coll_all_reduce(mg, R, out, (size_t)s * nbr_x * outw, R.coll)) return 1;     // aligned partial outputs of the ranks' giant shards
        R_HIP(R, hipEventRecord(R.ev_c, R.coll)); R_HIP(R, hipStreamWaitEvent(cs, R.ev_c, 0));
        R_CTX(R, sfg_reduce_rows_dev(ctx, out, (size_t)s * nbr_x * 2, L));
        return 0;
    };
    const int rc = exchange();
    i8_rotpre_free(pre8);                               // (the tile buffers stay in the context's pool for the next product)
    if (rc) coll_abort(mg, R);
    return rc;
}

// device-pointer form.  A_dev[i] / out_dev[i] belong to local rank i (device sfg_mgpu_ctx(mg, i)):
//   flags without SFG_TRANSPOSE (Q * X):   A_dev[i] = the whole [s][ceil(nrow / 8192)] input grid (replicated); out_dev[i] = [s][blk1 - blk0] of the rank's block columns
//   SFG_TRANSPOSE (Q' * X^T):              A_dev[i] = [s][blk1 - blk0] inputs of the rank's SNP blocks;          out_dev[i] = the whole [s][ceil(nrow / 8192)] result, on EVERY rank
// Stream-ordered on each rank's context queue; sfg_mgpu_synchronize waits.
extern "C" int sfg_mgpu_matmul_dev(sfg_mgpu *mg, const uint64_t *const *A_dev, int s, int in_level, int max_level, const sfg_mgeno *g, unsigned flags,
                                   uint64_t *const *out_dev) {
    MG_NEED(mg, mg != nullptr, "null engine");
    if (!g || g->shard.size() != mg->r.size()) MG_FAIL(mg, "sfg_mgpu_matmul: the matrix belongs to another engine");
    if (s < 1 || max_level < 1) MG_FAIL(mg, "sfg_mgpu_matmul: bad s / max_level");
    MG_NEED(mg, A_dev != nullptr && out_dev != nullptr, "null pointer tables");
    for (size_t i = 0; i < mg->r.size(); i++) {
        const bool has_in = (flags & SFG_TRANSPOSE) ? g->blk1[i] > g->blk0[i] : true, has_out = (flags & SFG_TRANSPOSE) ? true : g->blk1[i] > g->blk0[i];
        if ((has_in && has_out && !A_dev[i]) || (has_out && !out_dev[i])) MG_FAIL(mg, "sfg_mgpu_matmul_dev: null device pointer for local rank %zu", i);
    }
    MgApiScope scope(mg);
    return run_ranks(mg, [&](MgRank &R, int i) {
        R_HIP(R, hipSetDevice(R.device));
        if (flags & SFG_TRANSPOSE) return rank_contract(mg, R, i, A_dev[i], s, in_level, max_level, g, flags, out_dev[i]);
        if (g->shard[(size_t)i]) R_CTX(R, sfg_matmul_resident_dev(R.ctx, A_dev[i], s, in_level, max_level, g->shard[(size_t)i], flags & SFG_SQUARE, out_dev[i]));
        return 0;
    });
}

// host-pointer form = MatMult4StreamCompute on the sharded resident matrix (what the Go shim calls):
//   Q * X   : A_host [s][nbr][2][in_level+1][N] -> out_host [s][m_ct][2][max_level][N] (every block column, gathered from the local ranks; in a multi-process world
//             a process fills the block columns of ITS ranks and leaves the others untouched)
//   Q' * X^T: A_host [s][m_ct][...] (all SNP blocks; each rank takes its own) -> out_host [s][nbr][2][max_level][N], complete in every process
extern "C" int sfg_mgpu_matmul(sfg_mgpu *mg, const uint64_t *A_host, int s, int in_level, int max_level, const sfg_mgeno *g, unsigned flags, uint64_t *out_host) {
    MG_NEED(mg, mg != nullptr, "null engine");
    if (!g || g->shard.size() != mg->r.size()) MG_FAIL(mg, "sfg_mgpu_matmul: the matrix belongs to another engine");
    MG_NEED(mg, A_host != nullptr && out_host != nullptr, "null host buffers");
    MgApiScope scope(mg);                                // the I/O buffers below, the product's pools and the downloads are ONE call for the eviction rule of sfg_scratch
    const size_t N = SFG_N, ctw = 2 * (size_t)(in_level + 1) * N, outw = 2 * (size_t)max_level * N;
    const size_t nbr_x = (g->nrow + SFG_SLOTS - 1) / SFG_SLOTS, mct = (g->ncol + SFG_SLOTS - 1) / SFG_SLOTS;
    const bool tr = flags & SFG_TRANSPOSE;
    const size_t n = mg->r.size();
    std::vector<uint64_t *> A(n, nullptr), O(n, nullptr);
    int rc = run_ranks(mg, [&](MgRank &R, int i) {
        const size_t nloc = g->blk1[(size_t)i] - g->blk0[(size_t)i], na = tr ? nloc : nbr_x, no = tr ? nbr_x : nloc;
        R_CTX(R, sfg_scratch(R.ctx, "mg.Ain", std::max<size_t>(na, 1) * s * ctw * 8, (void **)&A[(size_t)i]));
        R_CTX(R, sfg_scratch(R.ctx, "mg.Oout", std::max<size_t>(no, 1) * s * outw * 8, (void **)&O[(size_t)i]));
        if (!tr) { if (na) R_CTX(R, sfg_memcpy_h2d(R.ctx, A[(size_t)i], A_host, (size_t)s * nbr_x * ctw * 8)); }
        else for (int r = 0; r < s && nloc; r++)      // row r of the input grid: the rank's block range
            R_CTX(R, sfg_memcpy_h2d(R.ctx, A[(size_t)i] + (size_t)r * nloc * ctw, A_host + ((size_t)r * mct + g->blk0[(size_t)i]) * ctw, nloc * ctw * 8));
        return 0;
    });
    if (rc) return rc;
    std::vector<const uint64_t *> Ac(A.begin(), A.end());
    rc = sfg_mgpu_matmul_dev(mg, Ac.data(), s, in_level, max_level, g, flags, O.data());
    if (rc) return rc;
    return run_ranks(mg, [&](MgRank &R, int i) {
        const size_t nloc = g->blk1[(size_t)i] - g->blk0[(size_t)i];
        if (tr) { if (i == 0) R_CTX(R, sfg_memcpy_d2h(R.ctx, out_host, O[(size_t)i], (size_t)s * nbr_x * outw * 8)); else R_CTX(R, sfg_ctx_synchronize(R.ctx)); }
        else for (int r = 0; r < s && nloc; r++)
            R_CTX(R, sfg_memcpy_d2h(R.ctx, out_host + ((size_t)r * mct + g->blk0[(size_t)i]) * outw, O[(size_t)i] + (size_t)r * nloc * outw, nloc * outw * 8));
        return 0;
    });
}

// ---------------------------------------------------------------- the association scan on G GPUs
// GenoBlockMult (gwas/assoc.go:340-420) hands the SNP batches of a chromosome file to assoc_num_blocks_parallel workers (:360-408); here the workers are the
// ranks: batch k goes to rank k % world, every rank streams ITS batches from the file (its own reader thread and pinned slots), multiplies them against its own
// copy of the call-wide rotation cache of `mat` (built once per rank: the rotations are a function of mat alone) and the outputs are copied to their
// positions of out_host [s][out_ct_capacity][2][max_level][N].  No collective: batches are independent.  In a multi-process world a process fills the
// positions of its ranks' batches and leaves the others untouched; *out_ct is the total in every process.
static int mgpu_assoc(sfg_mgpu *mg, int fmt, const char *path, size_t num_sample, size_t num_snp, const uint8_t *row_filter, const uint8_t *col_filter, size_t batch_snps,
                      const uint64_t *A_host, size_t nbr, int s, int in_level, int max_level, unsigned flags, uint64_t *out_host, size_t out_ct_capacity, size_t *out_ct,
                      double *sum_host, double *sqsum_host) {
    MG_NEED(mg, mg != nullptr, "null engine");
    if (s < 1 || !nbr || !out_host || !A_host || !path) MG_FAIL(mg, "sfg_mgpu_assoc: bad arguments");
    MgApiScope scope(mg);                                // (before the I/O buffers are requested: they belong to this call)
    const size_t N = SFG_N, ctw_in = 2 * (size_t)(in_level + 1) * N, ctw = 2 * (size_t)max_level * N;
    std::vector<size_t> totals(mg->r.size(), 0);
    const int rc = run_ranks(mg, [&](MgRank &R, int i) {
        uint64_t *A = nullptr, *out = nullptr;
        R_CTX(R, sfg_scratch(R.ctx, "mg.Ain", (size_t)s * nbr * ctw_in * 8, (void **)&A));
        R_CTX(R, sfg_scratch(R.ctx, "mg.assoc_out", (size_t)s * out_ct_capacity * ctw * 8, (void **)&out));
        R_CTX(R, sfg_memcpy_h2d(R.ctx, A, A_host, (size_t)s * nbr * ctw_in * 8));
        std::vector<std::pair<size_t, size_t>> ranges;
        R_CTX(R, assoc_stream_part(R.ctx, fmt, path, num_sample, num_snp, row_filter, col_filter, batch_snps, A, s, in_level, max_level, flags, out, out_ct_capacity, &totals[(size_t)i],
                                   sum_host, sqsum_host, R.rank, mg->world, &ranges));       // (sums: every rank writes the slices of its own batches - disjoint)
        for (const auto &rg : ranges) for (int r = 0; r < s; r++)
            R_CTX(R, sfg_memcpy_d2h(R.ctx, out_host + ((size_t)r * out_ct_capacity + rg.first) * ctw, out + ((size_t)r * out_ct_capacity + rg.first) * ctw, rg.second * ctw * 8));
        return 0;
    });
    if (rc) return rc;
    if (out_ct) *out_ct = totals[0];
    return 0;
}
extern "C" int sfg_mgpu_assoc_stream_bed(sfg_mgpu *mg, const char *bed_path, size_t num_sample, size_t num_snp, const uint8_t *row_filter, const uint8_t *col_filter,
                                         size_t batch_snps, const uint64_t *A_host, int s, int in_level, int max_level, unsigned flags,
                                         uint64_t *out_host, size_t out_ct_capacity, size_t *out_ct, double *sum_host, double *sqsum_host) {
    size_t nr = 0; for (size_t i = 0; i < num_sample; i++) nr += (!row_filter || row_filter[i]) ? 1 : 0;
    return mgpu_assoc(mg, 0, bed_path, num_sample, num_snp, row_filter, col_filter, batch_snps, A_host, (nr + SFG_SLOTS - 1) / SFG_SLOTS, s, in_level, max_level, flags,
                      out_host, out_ct_capacity, out_ct, sum_host, sqsum_host);
}
// kept_samples: the number of samples the row filter keeps (the file's sample count when row_filter is NULL) - it sizes `A_host` ([s][ceil(kept_samples / 8192)])
extern "C" int sfg_mgpu_assoc_stream_pgen(sfg_mgpu *mg, const char *pgen_path, const uint8_t *row_filter, const uint8_t *col_filter, size_t kept_samples,
                                          size_t batch_snps, const uint64_t *A_host, int s, int in_level, int max_level, unsigned flags,
                                          uint64_t *out_host, size_t out_ct_capacity, size_t *out_ct, double *sum_host, double *sqsum_host) {
    if (!kept_samples) MG_FAIL(mg, "sfg_mgpu_assoc_stream_pgen: kept_samples must be given (it sizes the input ciphertext grid)");
    return mgpu_assoc(mg, 1, pgen_path, 0, 0, row_filter, col_filter, batch_snps, A_host, (kept_samples + SFG_SLOTS - 1) / SFG_SLOTS, s, in_level, max_level, flags,
                      out_host, out_ct_capacity, out_ct, sum_host, sqsum_host);
}
