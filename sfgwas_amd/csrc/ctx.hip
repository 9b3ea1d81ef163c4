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

int sfg_encoder_init(sfg_ctx *ctx);      // encode.hip
void sfg_encoder_destroy(sfg_ctx *ctx);

extern "C" int sfg_ctx_create(sfg_ctx **out, int device, int logN, int nq, int np,
                              const uint64_t *moduli, const uint64_t *psi, double scale) {
    *out = nullptr;
    if (logN != SFG_LOGN) { g_create_error = "only logN = 14 (PN14QP438) is built into this library"; return 1; }
    if (nq < 1 || np < 1 || nq + np > SFG_MAXMOD) { g_create_error = "bad modulus counts"; return 1; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_create_error = "no HIP device: libsfgwas_hip has no CPU fallback"; return 1; }
    if (device < 0 || device >= ndev) { g_create_error = "bad device index"; return 1; }
    sfg_ctx *ctx = new sfg_ctx();
    ctx->device = device; ctx->nq = nq; ctx->np = np; ctx->nmod = nq + np; ctx->beta = (nq + np - 1) / np; ctx->scale = scale;
    auto fail = [&](const char *m) { g_create_error = m; sfg_ctx_destroy(ctx); return 1; };
    if (hipSetDevice(device) != hipSuccess) return fail("hipSetDevice failed");
    if (hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking) != hipSuccess) return fail("hipStreamCreate failed");
    ctx->stream = ctx->own_stream;
    {   // the auxiliary queue yields to the main one: its element-wise key-switch kernels fill gaps, they must not displace MAC workgroups
        int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
        if (hipStreamCreateWithPriority(&ctx->aux_stream, hipStreamNonBlocking, lo) != hipSuccess) return fail("hipStreamCreate failed");
    }
    for (int i = 0; i < 4; i++) if (hipEventCreateWithFlags(&ctx->ev_pipe[i], hipEventDisableTiming) != hipSuccess) return fail("hipEventCreate failed");
    ctx->pin_bytes = 64u << 20;
    if (hipHostMalloc((void **)&ctx->pin, ctx->pin_bytes, hipHostMallocDefault) != hipSuccess) return fail("hipHostMalloc failed");
    const int N = SFG_N;
    std::vector<double> twf((size_t)ctx->nmod * N), twi((size_t)ctx->nmod * N);
    for (int m = 0; m < ctx->nmod; m++) {
        u64 q = moduli[m];
        if (q >= (1ULL << 50) || (q - 1) % (2ULL * N)) return fail("modulus must be < 2^50 and == 1 mod 2N");
        ctx->q[m] = q;
        ctx->psi[m] = psi ? psi[m] : derive_psi(q, logN);
        if (h_powmod(ctx->psi[m], N, q) != q - 1) return fail("psi is not a primitive 2N-th root of unity");
        u64 psi_inv = h_invmod(ctx->psi[m], q), p = 1, pi = 1;
        for (int k = 0; k < N; k++) {
            uint32_t b = h_brev((uint32_t)k, logN);
            twf[(size_t)m * N + b] = (double)p;
            twi[(size_t)m * N + b] = (double)pi;
            p = h_mulmod(p, ctx->psi[m], q); pi = h_mulmod(pi, psi_inv, q);
        }
        ModConst &mc = ctx->modc_host[m];
        mc.q = (double)q; mc.qinv = 1.0 / (double)q; mc.qi = q;
        u64 ninv = h_invmod((u64)N, q);
        mc.ninv = (double)ninv; mc.ninv_q = (double)ninv / (double)q;
    }
    // late-stage twiddles (t = 8,4,2,1) of group p = j / 16: [T8, T4_0, T4_1, T2_0..3, T1_0..7, pad] laid out as
    // pack[p / 64][i < 8][p % 64] = {entry 2i, entry 2i+1}, so that a wave's 8 loads are 8 contiguous KiB
    auto build_pack = [&](const std::vector<double> &tw, std::vector<double2> &pk) {
        pk.assign((size_t)ctx->nmod * (N / 2), make_double2(0, 0));
        for (int m = 0; m < ctx->nmod; m++) {
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
    if (hipMalloc(&ctx->tw_fwd, twf.size() * sizeof(double)) != hipSuccess || hipMalloc(&ctx->tw_inv, twi.size() * sizeof(double)) != hipSuccess ||
        hipMalloc(&ctx->pack_fwd, pkf.size() * sizeof(double2)) != hipSuccess || hipMalloc(&ctx->pack_inv, pki.size() * sizeof(double2)) != hipSuccess ||
        hipMalloc(&ctx->modc, sizeof(ModConst) * SFG_MAXMOD) != hipSuccess) return fail("hipMalloc of tables failed");
    if (hipMemcpy(ctx->tw_fwd, twf.data(), twf.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(ctx->tw_inv, twi.data(), twi.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(ctx->pack_fwd, pkf.data(), pkf.size() * sizeof(double2), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(ctx->pack_inv, pki.data(), pki.size() * sizeof(double2), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(ctx->modc, ctx->modc_host, sizeof(ModConst) * SFG_MAXMOD, hipMemcpyHostToDevice) != hipSuccess) return fail("table upload failed");
    if (sfg_encoder_init(ctx)) { std::string e = ctx->err; return fail(e.c_str()); }
    *out = ctx;
    return 0;
}

extern "C" void sfg_ctx_destroy(sfg_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->own_stream) (void)hipStreamSynchronize(ctx->own_stream);
    if (ctx->aux_stream) (void)hipStreamSynchronize(ctx->aux_stream);
    for (auto &kv : ctx->rotkeys) { (void)hipFree(kv.second.key_dev); (void)hipFree(kv.second.index_dev); }
    sfg_phases_resolve(ctx);
    for (auto &kv : ctx->ksw_cache) (void)hipFree(kv.second);
    for (auto &kv : ctx->pool) (void)hipFree(kv.second.first);
    sfg_encoder_destroy(ctx);
    (void)hipFree(ctx->tw_fwd); (void)hipFree(ctx->tw_inv); (void)hipFree(ctx->pack_fwd); (void)hipFree(ctx->pack_inv); (void)hipFree(ctx->modc); (void)hipFree(ctx->ws); (void)hipFree(ctx->zeros_dev);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    if (ctx->aux_stream) (void)hipStreamDestroy(ctx->aux_stream);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    for (hipEvent_t e : ctx->ev_pool) (void)hipEventDestroy(e);
    for (int i = 0; i < 4; i++) if (ctx->ev_pipe[i]) (void)hipEventDestroy(ctx->ev_pipe[i]);
    delete ctx;
}

extern "C" const char *sfg_last_error(const sfg_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

extern "C" int sfg_ctx_synchronize(sfg_ctx *ctx) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    SFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
extern "C" int sfg_ctx_set_stream(sfg_ctx *ctx, void *s) { ctx->stream = s ? (hipStream_t)s : ctx->own_stream; return 0; }

int sfg_upload_small(sfg_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes) {
    if (!bytes) return 0;
    const size_t need = (bytes + 255) & ~(size_t)255;
    static const bool blocking = getenv("SFG_UPLOAD_BLOCKING") != nullptr;       // diagnostic switch
    if (blocking || need > ctx->pin_bytes / 4) {           // not "small": plain blocking copy, ordered after the stream
        SFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        SFG_HIP(ctx, hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
        return 0;
    }
    if (ctx->pin_head + need > ctx->pin_bytes) {           // wrap: earlier staged copies must have executed before their slots are reused
        SFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->aux_stream) SFG_HIP(ctx, hipStreamSynchronize(ctx->aux_stream));
        if (ctx->own_stream && ctx->own_stream != ctx->stream) SFG_HIP(ctx, hipStreamSynchronize(ctx->own_stream));
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
    SFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->ws) SFG_HIP(ctx, hipFree(ctx->ws));
    ctx->ws = nullptr; ctx->ws_bytes = 0;
    SFG_HIP(ctx, hipMalloc(&ctx->ws, bytes));
    ctx->ws_bytes = bytes;
    return 0;
}

extern "C" int sfg_malloc(sfg_ctx *ctx, void **p, size_t bytes) { SFG_HIP(ctx, hipSetDevice(ctx->device)); SFG_HIP(ctx, hipMalloc(p, bytes)); return 0; }
extern "C" int sfg_free(sfg_ctx *ctx, void *p) { SFG_HIP(ctx, hipSetDevice(ctx->device)); SFG_HIP(ctx, hipStreamSynchronize(ctx->stream)); SFG_HIP(ctx, hipFree(p)); return 0; }
extern "C" int sfg_memcpy_h2d(sfg_ctx *ctx, void *d, const void *s, size_t n) {
    SFG_HIP(ctx, hipSetDevice(ctx->device)); SFG_HIP(ctx, hipMemcpyAsync(d, s, n, hipMemcpyHostToDevice, ctx->stream)); SFG_HIP(ctx, hipStreamSynchronize(ctx->stream)); return 0;
}
extern "C" int sfg_memcpy_d2h(sfg_ctx *ctx, void *d, const void *s, size_t n) {
    SFG_HIP(ctx, hipSetDevice(ctx->device)); SFG_HIP(ctx, hipMemcpyAsync(d, s, n, hipMemcpyDeviceToHost, ctx->stream)); SFG_HIP(ctx, hipStreamSynchronize(ctx->stream)); return 0;
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
    if (e.second < bytes) {
        SFG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (e.first) SFG_HIP(ctx, hipFree(e.first));
        e.first = nullptr; e.second = 0;
        SFG_HIP(ctx, hipMalloc(&e.first, bytes));
        e.second = bytes;
    }
    *out = e.first;
    return 0;
}
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
extern "C" int sfg_ctx_has_rotkey(const sfg_ctx *ctx, uint64_t g) { return ctx->rotkeys.count(g) ? 1 : 0; }

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

extern "C" int sfg_ctx_load_rotkey(sfg_ctx *ctx, uint64_t g, const uint64_t *key_host, int mont) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    const int N = SFG_N; size_t words = (size_t)ctx->beta * 2 * ctx->nmod * N;
    RotKey rk;
    auto it = ctx->rotkeys.find(g);
    if (it != ctx->rotkeys.end()) rk = it->second;
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
    ctx->rotkeys[g] = rk;
    return 0;
}
