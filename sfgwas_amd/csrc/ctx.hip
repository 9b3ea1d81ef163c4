// ctx.hip — context, tables, memory and key management of libsfgwas_hip.
#include "common.hpp"

thread_local std::string g_create_error;

// smallest primitive root g of prime q, psi = g^((q-1)/2N): how lattigo's ring.NewRing derives its
// 2N-th root (used when the caller does not hand over ring.PsiMont)
static u64 derive_psi(u64 q, int logN) {
    u64 phi = q - 1, n = phi; std::vector<u64> fac;
    for (u64 p = 2; p * p <= n; p += (p == 2 ? 1 : 2)) if (n % p == 0) { fac.push_back(p); while (n % p == 0) n /= p; }
    if (n > 1) fac.push_back(n);
    for (u64 g = 2;; g++) {
        bool ok = true;
        for (u64 f : fac) if (h_powmod(g, phi / f, q) == 1) { ok = false; break; }
        if (ok) return h_powmod(g, phi >> (logN + 1), q);
    }
}

int sfg_encoder_init(sfg_ctx *ctx);      // encode.hip (tables into ctx->sh)
void sfg_encoder_destroy(SfgShared *sh);
int sfg_kernel_attrs_init(sfg_ctx *ctx);  // raises the dynamic-LDS limits of every big-LDS kernel on ctx->device (mac_dma/ntt/encode)
int mac_dma_set_attrs(sfg_ctx *ctx);
int mac_bc_set_attrs(sfg_ctx *ctx);
int ntt_set_attrs(sfg_ctx *ctx);
int encode_set_attrs(sfg_ctx *ctx);
int mac_i8_set_attrs(sfg_ctx *ctx);

// The configuration surface.
//   * What a deployment legitimately tunes is in the public sfg_config (include/sfgwas_hip.h; sfg_ctx_create_ex / sfg_mgpu_create_ex) and, for operators who cannot
//     change the caller, in ten environment variables read ONCE at context creation: SFG_MM_GROUP, SFG_MM_ACC_BUDGET_MB, SFG_ASSOC_ROTCACHE_MB, SFG_KSW_BUDGET_MB,
//     SFG_ENC_BATCH, SFG_UPLOAD_BLOCKING (here) and SFG_MGPU_TRANSPORT, SFG_MGPU_CACHE_GB, SFG_RCCL_LIB (mgpu.hip), plus the test switch itself.
//   * Three test-only switches exist in this build and are honoured only under SFG_ENABLE_TEST_HOOKS=1: SFG_TEST_SCRATCH_OOM, SFG_TEST_TIE_BAND_LOG2 and
//     SFG_MGPU_FORCE_COLLECTIVES (the exchange at world size 1).
//   * Every A/B and diagnostic switch of rounds 1 - 6 (which MAC kernel, which NTT form, queue schedules, CU masks, ...) and the superseded kernels they select exist
//     only in the A/B build (`make ab` -> sfgwas_amd/lib_ab/libsfgwas_hip.so, -DSFG_AB): a party process that inherits a stray variable from its shell cannot change
//     which kernels multiply.
static void apply_public_config(SfgConfig &c, const sfg_config *pc) {
    if (!pc || pc->struct_size < sizeof(uint32_t) * 2) return;
    const size_t have = pc->struct_size;
#define SFG_CFG_HAS(f) (have >= offsetof(sfg_config, f) + sizeof(pc->f))
    if (SFG_CFG_HAS(mm_group) && pc->mm_group > 0) { c.mm_group = pc->mm_group; c.mm_group_auto = false; }
    if (SFG_CFG_HAS(acc_budget_bytes) && pc->acc_budget_bytes) c.acc_budget = pc->acc_budget_bytes;
    if (SFG_CFG_HAS(assoc_rotcache_bytes) && pc->assoc_rotcache_bytes) c.assoc_cache_budget = pc->assoc_rotcache_bytes == SIZE_MAX ? 0 : pc->assoc_rotcache_bytes;
    if (SFG_CFG_HAS(ksw_budget_bytes) && pc->ksw_budget_bytes) c.ksw_budget = std::max<size_t>(pc->ksw_budget_bytes, 16ULL << 20);
    if (SFG_CFG_HAS(enc_batch) && pc->enc_batch > 0) c.enc_batch = std::min(std::max(pc->enc_batch, 64), 8192);
    if (SFG_CFG_HAS(upload_blocking) && pc->upload_blocking) c.upload_blocking = true;
#undef SFG_CFG_HAS
}
static void read_config(SfgConfig &c, const sfg_config *pc) {
    auto env = [](const char *n) { return getenv(n); };
    apply_public_config(c, pc);
    // ---- deployment: the environment overrides the caller's struct (an operator's last word)
    if (const char *e = env("SFG_MM_GROUP")) { c.mm_group = atoi(e); if (c.mm_group < 1) c.mm_group = 1; c.mm_group_auto = false; }
    if (const char *e = env("SFG_MM_ACC_BUDGET_MB")) c.acc_budget = (size_t)atoll(e) << 20;
    if (const char *e = env("SFG_ASSOC_ROTCACHE_MB")) c.assoc_cache_budget = (size_t)atoll(e) << 20;
    if (const char *e = env("SFG_KSW_BUDGET_MB")) { c.ksw_budget = (size_t)atoll(e) << 20; if (c.ksw_budget < (16ULL << 20)) c.ksw_budget = 16ULL << 20; }
    if (const char *e = env("SFG_ENC_BATCH")) { c.enc_batch = atoi(e); if (c.enc_batch < 64) c.enc_batch = 64; if (c.enc_batch > 8192) c.enc_batch = 8192; }
    if (env("SFG_UPLOAD_BLOCKING")) c.upload_blocking = true;
    // ---- tests
    if (const char *e = env("SFG_ENABLE_TEST_HOOKS")) c.test_hooks = atoi(e) == 1;
    if (c.test_hooks) {
        if (const char *e = env("SFG_TEST_SCRATCH_OOM")) c.test_scratch_oom = e;
        if (const char *e = env("SFG_TEST_TIE_BAND_LOG2")) c.tie_band = ldexp(1.0, atoi(e));       // a wider band sends ordinary coefficients through the exact re-derivation
    }
#ifdef SFG_AB
    // ---- the A/B build only: experiment switches (defaults = the measured configuration; results identical words unless a line says INVALID)
    if (const char *e = env("SFG_MAC_IMPL")) { c.mac_reg = !strcmp(e, "reg"); c.mac_bc = strcmp(e, "dma") != 0 && !c.mac_reg; c.mac_i8 = !strcmp(e, "i8"); }      // bc | dma | reg | i8
    if (const char *e = env("SFG_I8_KEEP_RESERVE_GB")) c.i8_keep_reserve = (size_t)atoll(e) << 30;
    if (const char *e = env("SFG_MAC_I8_BIG")) c.mac_i8_big = atoi(e) != 0;
    if (const char *e = env("SFG_MAC_I8_ROT")) { c.mac_i8_nolds = strcmp(e, "lds") != 0; c.mac_i8_ring = !strcmp(e, "ring"); }      // ring (default) | cache | lds
    if (env("SFG_MAC_I8_WG")) c.mac_i8_ring = false;
    if (const char *e = env("SFG_MAC_I8_DIAG")) { if (c.test_hooks) c.mac_i8_diag = atoi(e); }       // timing diagnostics with INVALID results
    if (const char *e = env("SFG_MAC_I8_WAVES")) c.mac_i8_waves = atoi(e) == 6 ? 6 : 12;
    if (const char *e = env("SFG_MAC_I8_STAGE")) c.stage_pack = atoi(e) != 0;
    if (const char *e = env("SFG_STAGE_GIANTS")) { c.stage_giants = atoi(e); if (c.stage_giants < 1) c.stage_giants = 1; if (c.stage_giants > 91) c.stage_giants = 91; }
    if (const char *e = env("SFG_STAGE_SAMEQ")) c.stage_same_queue = atoi(e) != 0;
    if (const char *e = env("SFG_MAC_I8_WG")) c.mac_i8_wg1 = atoi(e) == 1;
    if (const char *e = env("SFG_MAC_WC")) c.mac_wc = atoi(e) == 2 ? 2 : 1;
    if (const char *e = env("SFG_MM_OVERLAP")) c.no_overlap = atoi(e) == 0;
    if (env("SFG_MM_NO_OVERLAP")) c.no_overlap = true;
    if (const char *e = env("SFG_MM_ENC_OVERLAP")) c.no_enc_overlap = atoi(e) == 0;
    if (const char *e = env("SFG_NTT_HALF_IMPL")) c.ntt_half_full = !strcmp(e, "full");
    if (const char *e = env("SFG_NTT_FWD_IMPL")) c.ntt_fwd_full = !strcmp(e, "full");
    if (const char *e = env("SFG_MAC_PT")) c.mac_plain_pt = !strcmp(e, "plain");
    if (const char *e = env("SFG_CU_MAIN")) c.cu_main = e;
    if (const char *e = env("SFG_CU_ENC")) c.cu_enc = e;
    if (const char *e = env("SFG_CU_AUX")) c.cu_aux = e;
    if (const char *e = env("SFG_ASSOC_I8")) c.assoc_i8 = atoi(e) != 0;
    if (const char *e = env("SFG_I8_MOVER")) { c.i8_mover = atoi(e); if (c.i8_mover < 0) c.i8_mover = 0; c.i8_mover = c.i8_mover / 8 * 8; }
    if (const char *e = env("SFG_PT_RIDE")) { c.pt_ride = atoi(e); if (c.pt_ride < 0) c.pt_ride = 0; c.pt_ride = c.pt_ride / 8 * 8; }
    if (const char *e = env("SFG_PT_COMPACT")) c.pt_compact = atoi(e) != 0;
    if (const char *e = env("SFG_PT_KMAJOR")) c.pt_kmajor = atoi(e) != 0;
    if (const char *e = env("SFG_PT_RIDE_DEPTH")) { c.i8_mover_depth_ride = atoi(e); if (c.i8_mover_depth_ride < 1 || c.i8_mover_depth_ride > 3) c.i8_mover_depth_ride = 1; }
    if (const char *e = env("SFG_PT_RIDE_NT")) c.i8_mover_nt_ride = atoi(e) != 0;
    if (const char *e = env("SFG_I8_MOVER_DEPTH")) { c.i8_mover_depth = atoi(e); if (c.i8_mover_depth < 1 || c.i8_mover_depth > 3) c.i8_mover_depth = 3; }
#endif
}

// copy the shared scalars / table pointers into the context (read-only mirrors: the launch code reads ctx->q, ctx->modc, ...)
static void ctx_bind_shared(sfg_ctx *ctx, SfgShared *sh) {
    ctx->sh = sh; ctx->device = sh->device; ctx->logN = sh->logN; ctx->N = sh->N; ctx->nq = sh->nq; ctx->np = sh->np; ctx->nmod = sh->nmod;
    ctx->beta = sh->beta; ctx->scale = sh->scale; ctx->cfg = sh->cfg; ctx->test_hooks = sh->cfg.test_hooks;
    memcpy(ctx->q, sh->q, sizeof sh->q); memcpy(ctx->psi, sh->psi, sizeof sh->psi); memcpy(ctx->modc_host, sh->modc_host, sizeof sh->modc_host);
    ctx->tw_fwd = sh->tw_fwd; ctx->tw_inv = sh->tw_inv; ctx->pack_fwd = sh->pack_fwd; ctx->pack_inv = sh->pack_inv; ctx->modc = sh->modc;
}
#ifdef SFG_AB
// a queue restricted to the compute units named by `spec` ("lo-hi[,lo-hi...]", bits of hipExtStreamCreateWithCUMask); empty spec: nullptr (caller creates a plain queue)
static hipStream_t stream_with_cu_mask(const std::string &spec) {
    if (spec.empty()) return nullptr;
    uint32_t mask[16] = {0};                                   // up to 512 CUs
    const char *p = spec.c_str();
    while (*p) {
        char *e = nullptr; long lo = strtol(p, &e, 10), hi = lo + 1;
        if (e == p) return nullptr;
        if (*e == '-') { p = e + 1; hi = strtol(p, &e, 10); if (e == p) return nullptr; }
        for (long i = lo; i < hi && i < 512; i++) if (i >= 0) mask[i >> 5] |= 1u << (i & 31);
        p = *e == ',' ? e + 1 : e;
        if (*e && *e != ',') return nullptr;
    }
    hipStream_t st = nullptr;
    if (hipExtStreamCreateWithCUMask(&st, 16, mask) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return st;
}
#endif
// per-caller execution state: two queues, ordering events, the pinned staging ring
static const char *ctx_exec_init(sfg_ctx *ctx) {
    if (hipSetDevice(ctx->device) != hipSuccess) return "hipSetDevice failed";
    const SfgConfig &c = ctx->sh->cfg;
#ifdef SFG_AB
    if (!c.cu_main.empty() && !(ctx->own_stream = stream_with_cu_mask(c.cu_main))) return "SFG_CU_MAIN: bad CU list or hipExtStreamCreateWithCUMask failed";
    if (!c.cu_aux.empty() && !(ctx->aux_stream = stream_with_cu_mask(c.cu_aux))) return "SFG_CU_AUX: bad CU list or hipExtStreamCreateWithCUMask failed";
    if (!c.cu_enc.empty() && !(ctx->enc_stream = stream_with_cu_mask(c.cu_enc))) return "SFG_CU_ENC: bad CU list or hipExtStreamCreateWithCUMask failed";
#else
    (void)c;
#endif
    if (!ctx->own_stream && hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) return "hipStreamCreate failed";
    ctx->stream = ctx->own_stream;
    if (!ctx->aux_stream) {   // the auxiliary queue yields to the main one: its element-wise key-switch kernels fill gaps, they must not displace MAC workgroups
        int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&ctx->aux_stream, hipStreamNonBlocking, lo) != hipSuccess) return "hipStreamCreate failed";
    }
    if (!ctx->enc_stream && hipStreamCreateWithFlags(&ctx->enc_stream, hipStreamNonBlocking) != hipSuccess) return "hipStreamCreate failed";
    for (int i = 0; i < 4; i++) if (hipEventCreateWithFlags(&ctx->ev_pipe[i], hipEventDisableTiming) != hipSuccess) return "hipEventCreate failed";
    for (int i = 0; i < 4; i++) if (hipEventCreateWithFlags(&ctx->ev_enc[i], hipEventDisableTiming) != hipSuccess) return "hipEventCreate failed";
    ctx->pin_bytes = 64u << 20;
    if (hipHostMalloc((void **)&ctx->pin, ctx->pin_bytes, hipHostMallocDefault) != hipSuccess) return "hipHostMalloc failed";
    if (hipMalloc(&ctx->tie_count_dev, 32) != hipSuccess || hipMemset(ctx->tie_count_dev, 0, 32) != hipSuccess) return "hipMalloc failed";
    return nullptr;
}

extern "C" void sfg_config_default(sfg_config *c) { if (c) { memset(c, 0, sizeof *c); c->struct_size = (uint32_t)sizeof *c; } }
extern "C" int sfg_ctx_create(sfg_ctx **out, int device, int logN, int nq, int np,
                              const uint64_t *moduli, const uint64_t *psi, double scale) {
    return sfg_ctx_create_ex(out, device, logN, nq, np, moduli, psi, scale, nullptr);
}
extern "C" int sfg_ctx_create_ex(sfg_ctx **out, int device, int logN, int nq, int np,
                                 const uint64_t *moduli, const uint64_t *psi, double scale, const sfg_config *config) {
    if (!out) { g_create_error = "null result pointer"; return 1; }
    *out = nullptr;
    if (logN != SFG_LOGN) { g_create_error = "only logN = 14 (PN14QP438) is built into this library"; return 1; }
    if (nq < 1 || np < 1 || nq + np > SFG_MAXMOD) { g_create_error = "bad modulus counts"; return 1; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_create_error = "no HIP device: libsfgwas_hip has no CPU fallback"; return 1; }
    if (device < 0 || device >= ndev) { g_create_error = "bad device index"; return 1; }
    sfg_ctx *ctx = new sfg_ctx();
    SfgShared *sh = new SfgShared();
    ctx->sh = sh; ctx->device = device;
    sh->device = device; sh->nq = nq; sh->np = np; sh->nmod = nq + np; sh->beta = (nq + np - 1) / np; sh->scale = scale;
    read_config(sh->cfg, config);
    auto fail = [&](const char *m) { std::string msg = m; sfg_ctx_destroy(ctx); g_create_error = msg; return 1; };
    if (const char *e = ctx_exec_init(ctx)) return fail(e);
    const int N = SFG_N;
    std::vector<double> twf((size_t)sh->nmod * N), twi((size_t)sh->nmod * N);
    for (int m = 0; m < sh->nmod; m++) {
        u64 q = moduli[m];
        if (q >= (1ULL << 47) || (q - 1) % (2ULL * N)) return fail("modulus must be < 2^47 (exact fp64 arithmetic: 14 lazy NTT stages stay below 2^51) and == 1 mod 2N");
        sh->q[m] = q;
        // six signed base-256 digits hold a canonical plaintext word up to 0x7F7F7F7F7F7F: a ciphertext modulus beyond that (the top 0.8 % of the 47-bit range) keeps its MAC
        // on the fp64 kernel k_mac_bc<true>, whatever SFG_MAC_I8_BIG says
        if (m < nq && q > SFG_I8_BIG_QMAX) sh->cfg.mac_i8_big = false;
        sh->psi[m] = psi ? psi[m] : derive_psi(q, logN);
        if (h_powmod(sh->psi[m], N, q) != q - 1) return fail("psi is not a primitive 2N-th root of unity");
        u64 psi_inv = h_invmod(sh->psi[m], q), p = 1, pi = 1;
        for (int k = 0; k < N; k++) {
            uint32_t b = h_brev((uint32_t)k, logN);
            twf[(size_t)m * N + b] = (double)p;
            twi[(size_t)m * N + b] = (double)pi;
            p = h_mulmod(p, sh->psi[m], q); pi = h_mulmod(pi, psi_inv, q);
        }
        ModConst &mc = sh->modc_host[m];
        mc.q = (double)q; mc.qinv = 1.0 / (double)q; mc.qi = q;
        u64 ninv = h_invmod((u64)N, q);
        mc.ninv = (double)ninv; mc.ninv_q = (double)ninv / (double)q;
    }
    // late-stage twiddles (t = 8,4,2,1) of group p = j / 16: [T8, T4_0, T4_1, T2_0..3, T1_0..7, pad] laid out as
    // pack[p / 64][i < 8][p % 64] = {entry 2i, entry 2i+1}, so that a wave's 8 loads are 8 contiguous KiB
    auto build_pack = [&](const std::vector<double> &tw, std::vector<double2> &pk) {
        pk.assign((size_t)sh->nmod * (N / 2), make_double2(0, 0));
        for (int m = 0; m < sh->nmod; m++) {
            const double *t = tw.data() + (size_t)m * N;
            for (int p = 0; p < N / 16; p++) {
                double e[16]; int k = 0;
                e[k++] = t[1024 + p];
                for (int g = 0; g < 2; g++) e[k++] = t[2048 + 2 * p + g];
                for (int g = 0; g < 4; g++) e[k++] = t[4096 + 4 * p + g];
                for (int g = 0; g < 8; g++) e[k++] = t[8192 + 8 * p + g];
                e[k++] = 0.0;
                for (int i = 0; i < 8; i++) pk[(size_t)m * (N / 2) + (size_t)(p >> 6) * 512 + (size_t)i * 64 + (p & 63)] = make_double2(e[2 * i], e[2 * i + 1]);
            }
        }
    };
    std::vector<double2> pkf, pki; build_pack(twf, pkf); build_pack(twi, pki);
    if (hipMalloc(&sh->tw_fwd, twf.size() * sizeof(double)) != hipSuccess || hipMalloc(&sh->tw_inv, twi.size() * sizeof(double)) != hipSuccess ||
        hipMalloc(&sh->pack_fwd, pkf.size() * sizeof(double2)) != hipSuccess || hipMalloc(&sh->pack_inv, pki.size() * sizeof(double2)) != hipSuccess ||
        hipMalloc(&sh->modc, sizeof(ModConst) * SFG_MAXMOD) != hipSuccess || hipMalloc(&sh->zeros_dev, 512) != hipSuccess) return fail("hipMalloc of tables failed");
    if (hipMemcpy(sh->tw_fwd, twf.data(), twf.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(sh->tw_inv, twi.data(), twi.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(sh->pack_fwd, pkf.data(), pkf.size() * sizeof(double2), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(sh->pack_inv, pki.data(), pki.size() * sizeof(double2), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(sh->modc, sh->modc_host, sizeof(ModConst) * SFG_MAXMOD, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(sh->zeros_dev, 0, 512) != hipSuccess) return fail("table upload failed");
    {   // second 256 B: zero plaintext words in the packed-limb format (mac_dma.hip PACKED_ZERO)
        u64 pz[32]; for (int i = 0; i < 32; i++) pz[i] = PACKED_ZERO;
        if (hipMemcpy((char *)sh->zeros_dev + 256, pz, 256, hipMemcpyHostToDevice) != hipSuccess) return fail("table upload failed");
    }
    ctx_bind_shared(ctx, sh);
    if (sfg_encoder_init(ctx)) { std::string e = ctx->err; return fail(e.c_str()); }
    // dynamic-LDS limits are per (function, device): set here for this context's device, not behind process-wide flags
    if (mac_dma_set_attrs(ctx) || mac_bc_set_attrs(ctx) || ntt_set_attrs(ctx) || encode_set_attrs(ctx) || mac_i8_set_attrs(ctx)) { std::string e = ctx->err; return fail(e.c_str()); }
    *out = ctx;
    return 0;
}

// A second caller on the same key set: shares the immutable tables and keys, owns its queues / scratch / timers.
// Forks may run concurrently with each other and with the parent (one host thread per context at a time).  Keys must be
// loaded before concurrent use begins; destroy forks before (or after) the parent in any order - the shared part is
// released with the last of them.
extern "C" int sfg_ctx_fork(sfg_ctx *parent, sfg_ctx **out) {
    *out = nullptr;
    sfg_ctx *ctx = new sfg_ctx();
    ctx->is_fork = true;
    ctx_bind_shared(ctx, parent->sh);
    __atomic_add_fetch(&parent->sh->refs, 1, __ATOMIC_ACQ_REL);
    if (const char *e = ctx_exec_init(ctx)) { parent->err = std::string("sfg_ctx_fork: ") + e; sfg_ctx_destroy(ctx); return 1; }
    *out = ctx;
    return 0;
}

extern "C" void sfg_ctx_destroy(sfg_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
    if (ctx->aux_stream) (void)hipStreamSynchronize(ctx->aux_stream);
    if (ctx->enc_stream) (void)hipStreamSynchronize(ctx->enc_stream);
    if (ctx->user_stream) (void)hipStreamSynchronize(ctx->user_stream);
    sfg_phases_resolve(ctx);
    sfg_ptc_detach_all(ctx);           // matrices whose plaintext cache this context owns must not keep a pointer to it (their handles outlive a fork)
    for (auto &kv : ctx->ksw_cache) (void)hipFree(kv.second);
    for (auto &kv : ctx->host_pool) (void)hipHostFree(kv.second.first);
    for (auto &kv : ctx->pool) (void)hipFree(kv.second.first);
    (void)hipFree(ctx->ws); (void)hipFree(ctx->tie_count_dev);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
    if (ctx->enc_stream) (void)hipStreamDestroy(ctx->enc_stream);
    for (int i = 0; i < 4; i++) if (ctx->ev_enc[i]) (void)hipEventDestroy(ctx->ev_enc[i]);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    for (int i = 0; i < 4; i++) if (ctx->ev_pipe[i]) (void)hipEventDestroy(ctx->ev_pipe[i]);
    SfgShared *sh = ctx->sh;
    if (sh && __atomic_sub_fetch(&sh->refs, 1, __ATOMIC_ACQ_REL) == 0) {
        for (auto &kv : sh->rotkeys) { (void)hipFree(kv.second.key_dev); (void)hipFree(kv.second.index_dev); }
        sfg_encoder_destroy(sh);
        (void)hipFree(sh->tw_fwd); (void)hipFree(sh->tw_inv); (void)hipFree(sh->pack_fwd); (void)hipFree(sh->pack_inv); (void)hipFree(sh->modc); (void)hipFree(sh->zeros_dev); (void)hipFree(sh->sk_dev);
        delete sh;
    }
    delete ctx;
}

extern "C" const char *sfg_last_error(const sfg_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

extern "C" int sfg_ctx_synchronize(sfg_ctx *ctx) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    SFG_TRY(sfg_sync_all(ctx));                        // the main queue, the context's own queue and the auxiliary (key-switch) queue
    return sfg_encoder_check(ctx);
}
// An explicit stream only: the handle 0 is HIP's (and torch's) DEFAULT stream, which the library cannot be ordered against by accident - it used to
// mean "the context's own stream" and silently left collectives / torch ops unordered.  sfg_ctx_use_own_stream goes back to the private queue.
extern "C" int sfg_ctx_set_stream(sfg_ctx *ctx, void *s) {
    if (!s) SFG_FAIL(ctx, "sfg_ctx_set_stream: a NULL handle is HIP's default stream - create an explicit (non-default) stream, or call sfg_ctx_use_own_stream");
    ctx->user_stream = (hipStream_t)s; ctx->stream = ctx->main_stream(); return 0;
}
extern "C" int sfg_ctx_use_own_stream(sfg_ctx *ctx) { ctx->user_stream = nullptr; ctx->stream = ctx->main_stream(); return 0; }
int sfg_sync_all(sfg_ctx *ctx) {
    SFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->user_stream && ctx->user_stream != ctx->stream) SFG_HIP(ctx, hipStreamSynchronize(ctx->user_stream));
    if (ctx->own_stream && ctx->own_stream != ctx->stream) SFG_HIP(ctx, hipStreamSynchronize(ctx->own_stream));
    if (ctx->aux_stream && ctx->aux_stream != ctx->stream) SFG_HIP(ctx, hipStreamSynchronize(ctx->aux_stream));
    if (ctx->enc_stream && ctx->enc_stream != ctx->stream) SFG_HIP(ctx, hipStreamSynchronize(ctx->enc_stream));
    return 0;
}

int sfg_upload_small(sfg_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes) {
    if (!bytes) return 0;
    const size_t need = (bytes + 255) & ~(size_t)255;
    if (ctx->cfg.upload_blocking || need > ctx->pin_bytes / 4) {           // not "small": plain blocking copy, ordered after the stream
        SFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        SFG_HIP(ctx, hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
        return 0;
    }
    if (ctx->pin_head + need > ctx->pin_bytes) {           // wrap: earlier staged copies must have executed before their slots are reused
        SFG_TRY(sfg_sync_all(ctx));                        // both queues (and an installed external main stream) may hold staged copies
        ctx->pin_head = 0;
    }
    unsigned char *slot = ctx->pin + ctx->pin_head; ctx->pin_head += need;
    memcpy(slot, src_host, bytes);
    SFG_HIP(ctx, hipMemcpyAsync(dst_dev, slot, bytes, hipMemcpyHostToDevice, ctx->stream));
    return 0;
}
int sfg_stream_after(sfg_ctx *ctx, hipStream_t waiter, hipStream_t signaller) {
    if (waiter == signaller) return 0;
    if (ctx->ev_pool.size() < 64) { hipEvent_t e; SFG_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming)); ctx->ev_pool.push_back(e); ctx->ev_next = ctx->ev_pool.size() - 1; }
    hipEvent_t e = ctx->ev_pool[ctx->ev_next]; ctx->ev_next = (ctx->ev_next + 1) % ctx->ev_pool.size();
    SFG_HIP(ctx, hipEventRecord(e, signaller));            // a wait captures the record made here; re-recording later does not disturb it
    SFG_HIP(ctx, hipStreamWaitEvent(waiter, e, 0));
    return 0;
}

int sfg_ws_reserve(sfg_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->ws_bytes) return 0;
    SFG_TRY(sfg_sync_all(ctx));
    if (ctx->ws) SFG_HIP(ctx, hipFree(ctx->ws));
    ctx->ws = nullptr; ctx->ws_bytes = 0;
    SFG_HIP(ctx, hipMalloc(&ctx->ws, bytes));
    ctx->ws_bytes = bytes;
    return 0;
}

extern "C" int sfg_ctx_release_scratch(sfg_ctx *ctx);
// (a caller's buffer comes before the context's kept scratch - panels, accumulators, an association scan's rotation cache: when the device is full and no
//  library call is in progress, the pools are returned and the allocation tried once more)
extern "C" int sfg_malloc(sfg_ctx *ctx, void **p, size_t bytes) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    hipError_t err = hipMalloc(p, bytes);
    if (err == hipErrorOutOfMemory && ctx->api_depth == 0 && !ctx->pool.empty()) {
        (void)hipGetLastError();
        SFG_TRY(sfg_ctx_release_scratch(ctx));
        err = hipMalloc(p, bytes);
    }
    SFG_HIP(ctx, err);
    return 0;
}
extern "C" int sfg_free(sfg_ctx *ctx, void *p) { SFG_HIP(ctx, hipSetDevice(ctx->device)); SFG_HIP(ctx, hipStreamSynchronize(ctx->stream)); SFG_HIP(ctx, hipFree(p)); return 0; }
extern "C" int sfg_memcpy_h2d(sfg_ctx *ctx, void *d, const void *s, size_t n) {
    SFG_HIP(ctx, hipSetDevice(ctx->device)); SFG_HIP(ctx, hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, ctx->stream)); SFG_HIP(ctx, hipStreamSynchronize(ctx->stream)); return 0;
}
extern "C" int sfg_memcpy_d2h(sfg_ctx *ctx, void *d, const void *s, size_t n) {
    SFG_HIP(ctx, hipSetDevice(ctx->device)); SFG_HIP(ctx, hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, ctx->stream)); SFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return sfg_encoder_check(ctx);                     // results leave the device here: refuse while an unprovable encoder rounding is outstanding
}

void sfg_phases_resolve(sfg_ctx *ctx) {
    if (ctx->pending.empty()) return;
    (void)hipEventSynchronize(ctx->pending.back().e1);
    for (auto &p : ctx->pending) {
        float ms = 0; (void)hipEventSynchronize(p.e1); (void)hipEventElapsedTime(&ms, p.e0, p.e1);
        auto &st = ctx->phases[p.name]; st.ms += ms; st.launches += p.launches; st.bytes += p.bytes;
        (void)hipEventDestroy(p.e0); (void)hipEventDestroy(p.e1);
    }
    ctx->pending.clear();
}
int sfg_scratch(sfg_ctx *ctx, const char *name, size_t bytes, void **out) {
    auto &e = ctx->pool[name];
    ctx->pool_epoch[name] = ctx->api_epoch;
    bool inject = false;                                            // test switch: "the device is full" on the n-th request of one named buffer (tests/test_gpu_mgpu.py)
    if (ctx->test_hooks && !ctx->cfg.test_scratch_oom.empty() && ctx->api_depth > 0) {
        const std::string &t = ctx->cfg.test_scratch_oom; const size_t c = t.rfind(':');
        if (t.compare(0, c, name) == 0 && strlen(name) == (c == std::string::npos ? t.size() : c))
            inject = ++ctx->scratch_oom_seen == (c == std::string::npos ? 1 : atoi(t.c_str() + c + 1));
    }
    if (e.second < bytes || inject) {
        SFG_TRY(sfg_sync_all(ctx));
        if (e.first) SFG_HIP(ctx, hipFree(e.first));
        e.first = nullptr; e.second = 0;
        hipError_t err = inject ? hipErrorOutOfMemory : hipMalloc(&e.first, bytes);
        if (err != hipSuccess && ctx->api_depth > 0) {
            // The pools grow to the largest shape each buffer has served and are kept for the next call of that shape.  When the device is full, give back what
            // only EARLIER top-level calls asked for (nothing of the call in progress: every buffer it uses was requested under its epoch) and try once more.
            (void)hipGetLastError();
            bool freed_rot_copy = false;
            for (auto it = ctx->pool.begin(); it != ctx->pool.end();) {
                if (it->first != name && it->second.first && ctx->pool_epoch[it->first] < ctx->api_epoch) {
                    if (it->first.rfind("mi8.A", 0) == 0) freed_rot_copy = true;
                    (void)hipFree(it->second.first); ctx->pool_epoch.erase(it->first); it = ctx->pool.erase(it);
                } else ++it;
            }
            if (freed_rot_copy) { for (int b = 0; b < 2; b++) for (int i = 0; i < sfg_ctx::I8_SLOTS; i++) ctx->i8_slot[b][i] = sfg_ctx::I8Slot(); ctx->i8_gen++; }
            ctx->sp_shape = -1;
            auto &e2 = ctx->pool[name];
            err = hipMalloc(&e2.first, bytes);
            if (err != hipSuccess) { e2.first = nullptr; e2.second = 0; SFG_FAIL(ctx, "out of device memory: %zu bytes for scratch buffer '%s' (after returning the buffers of earlier calls)", bytes, name); }
            e2.second = bytes; *out = e2.first;
            return 0;
        }
        if (err != hipSuccess) { e.first = nullptr; SFG_FAIL(ctx, "out of device memory: %zu bytes for scratch buffer '%s' (sfg_ctx_release_scratch returns the grown pools)", bytes, name); }
        e.second = bytes;
    }
    *out = e.first;
    return 0;
}
// Returns every scratch buffer of the context to the device (after all of its queues have drained).  The pools grow to the largest shape a context has
// multiplied and are kept for the next call of that shape; a caller that moves to a very different shape beside a large resident matrix (kp = 15 products,
// then s = 2 products whose memory-chosen groups are larger) calls this in between instead of running out of HBM.
extern "C" int sfg_ctx_release_scratch(sfg_ctx *ctx) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    SFG_TRY(sfg_sync_all(ctx));
    for (auto &kv : ctx->pool) (void)hipFree(kv.second.first);
    ctx->pool.clear(); ctx->pool_epoch.clear();
    for (auto &kv : ctx->host_pool) (void)hipHostFree(kv.second.first);
    ctx->host_pool.clear();
    for (int b = 0; b < 2; b++) for (int i = 0; i < sfg_ctx::I8_SLOTS; i++) ctx->i8_slot[b][i] = sfg_ctx::I8Slot();      // the kept transposed rot copies lived in the pool
    ctx->i8_gen++; ctx->sp_shape = -1;
    return 0;
}
// pinned host scratch: a streamed scan reads its file through two slots of one batch each (1 GB at 500 000 samples x 8192 SNPs); pinning them costs ~0.2 s per
// GB and call, so they are kept like the device pools
int sfg_host_scratch(sfg_ctx *ctx, const char *name, size_t bytes, void **out) {
    auto &e = ctx->host_pool[name];
    if (e.second < bytes) {
        if (e.first) { SFG_TRY(sfg_sync_all(ctx)); SFG_HIP(ctx, hipHostFree(e.first)); }
        e.first = nullptr; e.second = 0;
        SFG_HIP(ctx, hipHostMalloc(&e.first, bytes, hipHostMallocDefault));
        e.second = bytes;
    }
    *out = e.first;
    return 0;
}
extern "C" int sfg_ctx_scratch_bytes(const sfg_ctx *ctx, const char *prefix, size_t *bytes) {
    size_t n = 0; const std::string pre = prefix ? prefix : "";
    for (const auto &kv : ctx->pool) if (kv.second.first && kv.first.rfind(pre, 0) == 0) n += kv.second.second;
    if (bytes) *bytes = n;
    return 0;
}
// The int8 MAC keeps a transposed copy of a rotation-cache operand, keyed by its address, shape and a per-context generation counter that every library
// writer of rotation rows advances.  A caller that writes a cache buffer by any other route (an RCCL all-gather straight into the layout, a device copy of a
// saved cache, another context's build) tells the multiplying context so with this call before the next *_rc_dev product.
extern "C" int sfg_rotcache_invalidate(sfg_ctx *ctx) { ctx->i8_gen++; return 0; }
extern "C" int sfg_ctx_clear_phases(sfg_ctx *ctx) { sfg_phases_resolve(ctx); ctx->phases.clear(); return 0; }
extern "C" double sfg_last_phase_ms(const sfg_ctx *ctx, const char *phase) {
    sfg_phases_resolve(const_cast<sfg_ctx *>(ctx));
    auto it = ctx->phases.find(phase); return it == ctx->phases.end() ? -1.0 : it->second.ms;
}
extern "C" double sfg_last_phase_bytes(const sfg_ctx *ctx, const char *phase) {
    sfg_phases_resolve(const_cast<sfg_ctx *>(ctx));
    auto it = ctx->phases.find(phase); return it == ctx->phases.end() ? -1.0 : it->second.bytes;
}
extern "C" int sfg_last_phase_launches(const sfg_ctx *ctx, const char *phase) {
    sfg_phases_resolve(const_cast<sfg_ctx *>(ctx));
    auto it = ctx->phases.find(phase); return it == ctx->phases.end() ? -1 : it->second.launches;
}

// ---------------------------------------------------------------- rotation keys
extern "C" uint64_t sfg_galois_for_rotation(const sfg_ctx *ctx, int k) {
    const u64 M = 2ULL * SFG_N; int n = SFG_SLOTS; k %= n; if (k < 0) k += n;
    u64 g = 1; for (int i = 0; i < k; i++) g = (g * 5) % M; return g;
}
extern "C" int sfg_ctx_has_rotkey(const sfg_ctx *ctx, uint64_t g) { return ctx->rotkeys().count(g) ? 1 : 0; }

extern "C" int sfg_ctx_export_rotkey(sfg_ctx *ctx, uint64_t g, uint64_t *key_host) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    auto it = ctx->rotkeys().find(g);
    if (it == ctx->rotkeys().end()) SFG_FAIL(ctx, "export_rotkey: no key loaded for galois element %llu", (unsigned long long)g);
    const size_t words = (size_t)ctx->beta * 2 * ctx->nmod * SFG_N;
    SFG_HIP(ctx, hipMemcpyAsync(key_host, it->second.key_dev, words * 8, hipMemcpyDeviceToHost, ctx->stream));
    SFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// lattigo ring.InvMForm: x * 2^-64 mod q, applied when the caller hands keys in Montgomery form
__global__ void k_from_montgomery(u64 *rows, int nmod, const ModConst *modc, size_t total_rows) {
    size_t row = blockIdx.y; int m = (int)(row % nmod); u64 q = modc[m].qi;
    // 2^-64 mod q computed per block (setup path, not hot)
    __shared__ u64 r64inv;
    if (threadIdx.x == 0) {
        u64 r = ((~0ULL) % q + 1) % q;          // 2^64 mod q
        // modular inverse by Fermat
        u64 e = q - 2, base = r, acc = 1;
        while (e) { if (e & 1) acc = d_mulmod_u64(acc, base, q); base = d_mulmod_u64(base, base, q); e >>= 1; }
        r64inv = acc;
    }
    __syncthreads();
    u64 *p = rows + row * SFG_N;
    for (int x = blockIdx.x * blockDim.x + threadIdx.x; x < SFG_N; x += gridDim.x * blockDim.x) p[x] = d_mulmod_u64(p[x], r64inv, q);
}

// cryptoParams.Sk.Value (crypto.go:44): this party's secret-key shard, the Q rows [nq][N] in the NTT domain
extern "C" int sfg_ctx_load_secret_key(sfg_ctx *ctx, const uint64_t *sk_host, int mont) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    const size_t words = (size_t)ctx->nq * SFG_N;
    if (!ctx->sh->sk_dev) SFG_HIP(ctx, hipMalloc(&ctx->sh->sk_dev, words * 8));
    SFG_HIP(ctx, hipMemcpyAsync(ctx->sh->sk_dev, sk_host, words * 8, hipMemcpyHostToDevice, ctx->stream));
    if (mont) {
        dim3 grid(8, (unsigned)ctx->nq);
        hipLaunchKernelGGL(k_from_montgomery, grid, dim3(256), 0, ctx->stream, ctx->sh->sk_dev, ctx->nq, ctx->modc, (size_t)grid.y);
    }
    SFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int sfg_ctx_load_rotkey(sfg_ctx *ctx, uint64_t g, const uint64_t *key_host, int mont) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    const int N = SFG_N; size_t words = (size_t)ctx->beta * 2 * ctx->nmod * N;
    RotKey rk;
    auto it = ctx->rotkeys().find(g);
    if (it != ctx->rotkeys().end()) rk = it->second;
    else { SFG_HIP(ctx, hipMalloc(&rk.key_dev, words * 8)); SFG_HIP(ctx, hipMalloc(&rk.index_dev, N * sizeof(uint16_t))); }
    SFG_HIP(ctx, hipMemcpyAsync(rk.key_dev, key_host, words * 8, hipMemcpyHostToDevice, ctx->stream));
    if (mont) {
        dim3 grid(8, (unsigned)((size_t)ctx->beta * 2 * ctx->nmod));
        hipLaunchKernelGGL(k_from_montgomery, grid, dim3(256), 0, ctx->stream, rk.key_dev, ctx->nmod, ctx->modc, (size_t)grid.y);
    }
    // lattigo ring.PermuteNTTIndex: out[i] = in[index[i]]
    std::vector<uint16_t> idx(N); u64 mask = 2ULL * N - 1;
    for (int i = 0; i < N; i++) { u64 t1 = 2ULL * h_brev((uint32_t)i, SFG_LOGN) + 1; u64 t2 = ((g * t1 & mask) - 1) >> 1; idx[i] = (uint16_t)h_brev((uint32_t)t2, SFG_LOGN); }
    SFG_HIP(ctx, hipMemcpyAsync(rk.index_dev, idx.data(), N * sizeof(uint16_t), hipMemcpyHostToDevice, ctx->stream));
    SFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->rotkeys()[g] = rk;
    return 0;
}
