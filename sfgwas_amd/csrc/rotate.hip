// rotate.hip — ciphertext slot rotations: crypto.RotateRightWithEvaluator / RotateRight
// (crypto/basics.go:201-224) -> lattigo ckks.Evaluator.RotateNew = hybrid key switch of c1 with the Galois key,
// add to c0, NTT-domain automorphism of both polynomials.  Used for the baby-step rotation cache
// (gwas/matmult.go:1114,1375) and the giant-step alignment (:1207,1476).
//
// Batched over ciphertexts (each may use a different key).  Per ciphertext at level l (nl = l+1 moduli,
// alpha = np special primes per digit, beta = ceil(nl/alpha) digits, nt = nl + np targets):
//   1. c2 = INTT(c1)                                   nl rows
//   2. digit i -> exact (float-corrected) basis extension to every target outside the digit     (k_ksw_extend)
//      (inside the digit the original NTT rows are reused), then NTT of the extended rows         beta*nt rows
//   3. acc_{0,1}[t] = sum_i ext[i][t] * key[i][{0,1}][t]                                           (k_ksw_inner)
//   4. ModDown: INTT of the np special-prime rows, basis extension P -> Q, NTT                    (k_moddown_extend)
//   5. out0 = perm(c0 + (acc0 - ext0) / P), out1 = perm((acc1 - ext1) / P)                        (k_ksw_finish)
// The float correction v = uint64(sum float64(y_m)/float64(q_m)) is lattigo's (ring.Decomposer /
// FastBasisExtender); the oracle mirrors it, so results agree bit for bit.
#include "common.hpp"
#include "kernels.hpp"

constexpr int KSW_MAXA = 4;       // max primes per digit (= max np)
constexpr int KSW_MAXDIG = 8;

struct ExtConst {
    int a;                                        // source moduli in this digit at this level
    int src[KSW_MAXA];                            // their global modulus indices
    double qhat_inv[KSW_MAXA], qhat_inv_q[KSW_MAXA];          // (D/q_m)^-1 mod q_m, and that / q_m
    double qhat_t[SFG_MAXMOD][KSW_MAXA], qhat_t_q[SFG_MAXMOD][KSW_MAXA];   // (D/q_m) mod q_t, / q_t  (t = global modulus index)
    double D_t[SFG_MAXMOD], D_t_q[SFG_MAXMOD];    // D mod q_t, / q_t
};
struct KswConst {
    int level, nl, np, nt, beta, alpha;
    int tmod[SFG_MAXMOD];                         // target slot -> global modulus index (Q_0..level then P)
    int digit_of[SFG_MAXMOD];                     // target slot -> digit that contains it (or -1 for P targets)
    ExtConst dig[KSW_MAXDIG];
    ExtConst pq;                                  // special primes -> Q (ModDown)
    double pinv[SFG_MAXMOD], pinv_q[SFG_MAXMOD];  // P^-1 mod q_t by global modulus index
};

static void fill_ext(const sfg_ctx *ctx, ExtConst &e, const std::vector<int> &src) {
    memset(&e, 0, sizeof e);
    e.a = (int)src.size();
    for (int m = 0; m < e.a; m++) e.src[m] = src[m];
    for (int m = 0; m < e.a; m++) {
        u64 qm = ctx->q[src[m]], h = 1;
        for (int k = 0; k < e.a; k++) if (k != m) h = h_mulmod(h, ctx->q[src[k]] % qm, qm);
        u64 hi = h_invmod(h, qm);
        e.qhat_inv[m] = (double)hi; e.qhat_inv_q[m] = (double)hi / (double)qm;
    }
    for (int t = 0; t < ctx->nmod; t++) {
        u64 qt = ctx->q[t], D = 1 % qt;
        for (int m = 0; m < e.a; m++) {
            u64 ht = 1 % qt;
            for (int k = 0; k < e.a; k++) if (k != m) ht = h_mulmod(ht, ctx->q[src[k]] % qt, qt);
            e.qhat_t[t][m] = (double)ht; e.qhat_t_q[t][m] = (double)ht / (double)qt;
            D = h_mulmod(D, ctx->q[src[m]] % qt, qt);
        }
        e.D_t[t] = (double)D; e.D_t_q[t] = (double)D / (double)qt;
    }
}


static int get_ksw(sfg_ctx *ctx, int level, KswConst **dev, KswConst *host) {
    KswConst kc; memset(&kc, 0, sizeof kc);
    kc.level = level; kc.nl = level + 1; kc.np = ctx->np; kc.alpha = ctx->np; kc.nt = kc.nl + kc.np;
    kc.beta = (kc.nl + kc.alpha - 1) / kc.alpha;
    if (kc.alpha > KSW_MAXA || kc.beta > KSW_MAXDIG) SFG_FAIL(ctx, "key-switch shape unsupported (alpha > 4 or beta > 8)");
    if (kc.beta * kc.nt > SFG_MAXPATTERN) SFG_FAIL(ctx, "key-switch shape unsupported (beta * (level + 1 + np) = %d rows exceed the modulus pattern table)", kc.beta * kc.nt);
    for (int t = 0; t < kc.nt; t++) { kc.tmod[t] = t < kc.nl ? t : ctx->nq + (t - kc.nl); kc.digit_of[t] = t < kc.nl ? t / kc.alpha : -1; }
    for (int i = 0; i < kc.beta; i++) {
        std::vector<int> src;
        for (int m = i * kc.alpha; m < (i + 1) * kc.alpha && m < kc.nl; m++) src.push_back(m);
        fill_ext(ctx, kc.dig[i], src);
    }
    std::vector<int> ps; for (int p = 0; p < kc.np; p++) ps.push_back(ctx->nq + p);
    fill_ext(ctx, kc.pq, ps);
    for (int t = 0; t < kc.nl; t++) {
        u64 qt = ctx->q[t], P = 1;
        for (int p = 0; p < kc.np; p++) P = h_mulmod(P, ctx->q[ctx->nq + p] % qt, qt);
        u64 pi = h_invmod(P, qt);
        kc.pinv[t] = (double)pi; kc.pinv_q[t] = (double)pi / (double)qt;
    }
    *host = kc;
    auto it = ctx->ksw_cache.find(level);
    if (it == ctx->ksw_cache.end()) {
        KswConst *d = nullptr;
        SFG_HIP(ctx, hipMalloc(&d, sizeof(KswConst)));
        SFG_HIP(ctx, hipMemcpy(d, &kc, sizeof(KswConst), hipMemcpyHostToDevice));
        ctx->ksw_cache[level] = d; *dev = d;
    } else *dev = (KswConst *)it->second;
    return 0;
}

// general modular product of two canonical residues held in fp64 (both variable): result in (-q, q)
__device__ __forceinline__ double mulmod2(double a, double b, double q, double qinv) {
    double h = a * b;
    double l = __builtin_fma(a, b, -h);
    double qh = __builtin_rint(h * qinv);
    double r = __builtin_fma(-qh, q, h);
    return r + l;
}

// y_m, v and the extension to one target modulus (lattigo reconstructRNS + multSum restated)
__device__ __forceinline__ void ext_prepare(const ExtConst &e, const ModConst *modc, const double (&x)[KSW_MAXA], double (&y)[KSW_MAXA], double &v) {
    double vf = 0.0;
#pragma unroll
    for (int m = 0; m < KSW_MAXA; m++) {
        if (m < e.a) {
            const ModConst mc = modc[e.src[m]];
            y[m] = canon(mulmod_lazy(x[m], e.qhat_inv[m], e.qhat_inv_q[m], mc.q), mc.q, mc.qinv);
            vf += y[m] / mc.q;                                  // IEEE division, accumulated in modulus order
        }
    }
    v = (double)(u64)vf;
}
__device__ __forceinline__ double ext_target(const ExtConst &e, int tg, double qt, double qtinv, const double (&y)[KSW_MAXA], double v) {
    double acc = -mulmod_lazy(v, e.D_t[tg], e.D_t_q[tg], qt);
#pragma unroll
    for (int m = 0; m < KSW_MAXA; m++) if (m < e.a) acc += mulmod_lazy(y[m], e.qhat_t[tg][m], e.qhat_t_q[tg][m], qt);
    return canon(acc, qt, qtinv);
}

// grid (N/256, beta, B). c2: [B][nl][N] coefficient-domain c1; cx: original ct (c1 rows at +nl*N); ext: [B][beta][nt][N]
__global__ void __launch_bounds__(256) k_ksw_extend(const u64 *c2, const u64 *ct_in, u64 *ext, const KswConst *kcp, const ModConst *modc) {
    const KswConst &kc = *kcp;
    const int N = SFG_N, x = blockIdx.x * 256 + threadIdx.x, i = blockIdx.y; const size_t b = blockIdx.z;
    const ExtConst &e = kc.dig[i];
    const u64 *c2b = c2 + b * (size_t)kc.nl * N;
    u64 *eb = ext + (b * kc.beta + i) * (size_t)kc.nt * N;
    double xs[KSW_MAXA], y[KSW_MAXA], v = 0.0;
#pragma unroll
    for (int m = 0; m < KSW_MAXA; m++) xs[m] = m < e.a ? u64_to_f64(c2b[(size_t)e.src[m] * N + x]) : 0.0;
    if (e.a > 1) ext_prepare(e, modc, xs, y, v);
    for (int t = 0; t < kc.nt; t++) {
        const int tg = kc.tmod[t];
        u64 outv;
        if (kc.digit_of[t] == i) continue;                                            // in-digit: the original NTT row, which k_ksw_inner reads where it lies
        else if (e.a == 1) outv = f64_to_u64(canon(xs[0], modc[tg].q, modc[tg].qinv)); // single-prime digit: raw copy mod q_t
        else outv = f64_to_u64(ext_target(e, tg, modc[tg].q, modc[tg].qinv, y, v));
        eb[(size_t)t * N + x] = outv;
    }
}

// grid (N/512, nt, B). acc: [B][2][nt][N]; inidx[b] = which decomposed input feeds output b.  Two coefficients per thread: 16-byte loads and stores.
// The row of digit i at a target inside digit i is the input's own NTT row (polynomial 1 of ct_in): read there, never copied (round 4: 15 of ~250 row transfers per switch less)
__global__ void __launch_bounds__(256) k_ksw_inner(const u64 *ext, const u64 *ct_in, const u64 *const *keys, const int *inidx, u64 *acc, const KswConst *kcp, const ModConst *modc, int nmod) {
    const KswConst &kc = *kcp;
    const int N = SFG_N, x = 2 * (blockIdx.x * 256 + threadIdx.x), t = blockIdx.y; const size_t b = blockIdx.z;
    const int tg = kc.tmod[t];
    const double q = modc[tg].q, qinv = modc[tg].qinv;
    const u64 *key = keys[b];
    const size_t bi = (size_t)inidx[b];
    double a0x = 0.0, a0y = 0.0, a1x = 0.0, a1y = 0.0;
    for (int i = 0; i < kc.beta; i++) {
        const ulonglong2 ev = kc.digit_of[t] == i ? *reinterpret_cast<const ulonglong2 *>(ct_in + (bi * 2 * (size_t)kc.nl + kc.nl + t) * N + x)
                                                  : *reinterpret_cast<const ulonglong2 *>(ext + ((bi * kc.beta + i) * (size_t)kc.nt + t) * N + x);
        const ulonglong2 k0 = *reinterpret_cast<const ulonglong2 *>(key + (((size_t)i * 2 + 0) * nmod + tg) * N + x);
        const ulonglong2 k1 = *reinterpret_cast<const ulonglong2 *>(key + (((size_t)i * 2 + 1) * nmod + tg) * N + x);
        const double ex = u64_to_f64(ev.x), ey = u64_to_f64(ev.y);
        a0x += mulmod2(ex, u64_to_f64(k0.x), q, qinv); a0y += mulmod2(ey, u64_to_f64(k0.y), q, qinv);
        a1x += mulmod2(ex, u64_to_f64(k1.x), q, qinv); a1y += mulmod2(ey, u64_to_f64(k1.y), q, qinv);
    }
    *reinterpret_cast<ulonglong2 *>(acc + ((b * 2 + 0) * (size_t)kc.nt + t) * N + x) = make_ulonglong2(f64_to_u64(canon(a0x, q, qinv)), f64_to_u64(canon(a0y, q, qinv)));
    *reinterpret_cast<ulonglong2 *>(acc + ((b * 2 + 1) * (size_t)kc.nt + t) * N + x) = make_ulonglong2(f64_to_u64(canon(a1x, q, qinv)), f64_to_u64(canon(a1y, q, qinv)));
}

// grid (N/256, 2, B): special-prime rows of acc (already INTT'd) -> ext2 [B][2][nl][N] coefficient domain
__global__ void __launch_bounds__(256) k_moddown_extend(const u64 *acc, u64 *ext2, const KswConst *kcp, const ModConst *modc) {
    const KswConst &kc = *kcp;
    const int N = SFG_N, x = blockIdx.x * 256 + threadIdx.x, p = blockIdx.y; const size_t b = blockIdx.z;
    const ExtConst &e = kc.pq;
    const u64 *ap = acc + ((b * 2 + p) * (size_t)kc.nt + kc.nl) * N;
    double xs[KSW_MAXA], y[KSW_MAXA], v = 0.0;
#pragma unroll
    for (int m = 0; m < KSW_MAXA; m++) xs[m] = m < e.a ? u64_to_f64(ap[(size_t)m * N + x]) : 0.0;
    if (e.a > 1) ext_prepare(e, modc, xs, y, v);
    for (int t = 0; t < kc.nl; t++) {
        u64 outv = e.a == 1 ? f64_to_u64(canon(xs[0], modc[t].q, modc[t].qinv)) : f64_to_u64(ext_target(e, t, modc[t].q, modc[t].qinv, y, v));
        ext2[((b * 2 + p) * (size_t)kc.nl + t) * N + x] = outv;
    }
}

// Optional output form of a key-switch job: instead of a u64 ciphertext, the fp64 operand rows the MAC reads (mac_dma.hip k_rot_to_f64: one centred double per
// word for a small modulus, a signed {lo, hi} pair for the 46-bit one; only the first L moduli, which are all the MAC accumulates over).  The baby-step
// rotation cache is written in this form directly: no u64 copy of it exists, and the rows of modulus L are never produced.
struct F64Form { int on, L, centre; size_t rowf; int plane_of[SFG_MAXMOD], big[SFG_MAXMOD]; };
__device__ __forceinline__ void store_rot_f64(double *row, const F64Form &ff, int t, int x, double v, double q) {     // v canonical in [0, q)
    const int N = SFG_N;
    double *o = row + (size_t)ff.plane_of[t] * N;
    if (ff.big[t]) {
        const double wc = v > __builtin_floor(q * 0.5) ? v - q : v;                         // centred; q odd: floor(q/2) = (q-1)/2 = q >> 1
        const double hi = __builtin_floor((wc + 4194304.0) * 0x1p-23), lo = wc - hi * 8388608.0;
        o[2 * x] = lo; o[2 * x + 1] = hi;
    } else o[x] = (ff.centre && v > __builtin_floor(q * 0.5)) ? v - q : v;
}
// grid (N/256, nl, B): out = perm(c0 + (acc0 - ext0)/P), perm((acc1 - ext1)/P); out[x] = in[index[x]]
// add1 (nullable): [nin][nl][N] rows added to polynomial 1 (relinearisation: the degree-1 term of the tensor product)
__global__ void __launch_bounds__(256) k_ksw_finish(const u64 *ct_in, const int *inidx, const u64 *acc, const u64 *ext2, const uint16_t *const *index,
                                                   u64 *const *ct_out, const KswConst *kcp, const ModConst *modc, const u64 *add1, F64Form ff) {
    const KswConst &kc = *kcp;
    const int N = SFG_N, x = blockIdx.x * 256 + threadIdx.x, t = blockIdx.y; const size_t b = blockIdx.z;
    if (ff.on && t >= ff.L) return;
    const double q = modc[t].q, qinv = modc[t].qinv, pinv = kc.pinv[t], pinv_q = kc.pinv_q[t];
    const int src = index[b][x];
#pragma unroll
    for (int p = 0; p < 2; p++) {
        const double a = u64_to_f64(acc[((b * 2 + p) * (size_t)kc.nt + t) * N + src]);
        const double e = u64_to_f64(ext2[((b * 2 + p) * (size_t)kc.nl + t) * N + src]);
        double r = mulmod_lazy(a - e, pinv, pinv_q, q);
        if (p == 0) r += u64_to_f64(ct_in[((size_t)inidx[b] * 2 * (size_t)kc.nl + t) * N + src]);
        else if (add1) r += u64_to_f64(add1[((size_t)inidx[b] * (size_t)kc.nl + t) * N + src]);
        const double v = canon(r, q, qinv);
        if (ff.on) store_rot_f64(reinterpret_cast<double *>(ct_out[b]) + (size_t)p * ff.rowf, ff, t, x, v, q);
        else ct_out[b][((size_t)p * kc.nl + t) * N + x] = f64_to_u64(v);
    }
}

__global__ void __launch_bounds__(256) k_ct_add(const u64 *a, const u64 *b, u64 *out, int nl, const ModConst *modc, size_t total_rows) {
    const int N = SFG_N; const size_t row = blockIdx.x / (N / 256); const int m = (int)(row % nl);
    const u64 q = modc[m].qi;
    const size_t off = row * N + (blockIdx.x % (N / 256)) * 256 + threadIdx.x;
    u64 v = a[off] + b[off]; out[off] = v >= q ? v - q : v;
}

// Key-switch a batch of jobs.  `in` holds nin ciphertext-shaped inputs [nin][2][nl][N] whose polynomial 1 is switched;
// job k reads input job_in[k], uses key keyp[k] and automorphism table idxp[k], and writes
//   out0 = perm(in.p0 + d0), out1 = perm(d1 [+ add1[in]])            to outp[k].
// Decomposition (steps 1-2) is done once per INPUT and shared by all its jobs ("hoisting"): the per-key work is
// only the inner product, ModDown and the automorphism.  Same arithmetic, same bits.
static int launch_keyswitch_jobs(sfg_ctx *ctx, const u64 *in, int nin, int level, const std::vector<int> &job_in, const std::vector<const u64 *> &keyp,
                                 const std::vector<const uint16_t *> &idxp, const std::vector<u64 *> &outp, const u64 *add1, const F64Form *ffp = nullptr) {
    const int N = SFG_N, nl = level + 1;
    F64Form ff; memset(&ff, 0, sizeof ff); if (ffp) ff = *ffp;
    KswConst *kcd; KswConst kc;
    SFG_TRY(get_ksw(ctx, level, &kcd, &kc));
    const size_t ctw = (size_t)2 * nl * N;
    const int nr = (int)job_in.size();
    if (!nr) return 0;
    // inputs are processed in groups whose decomposition fits the budget (4 GiB by default); jobs of a group in chunks whose acc/ext2 fit it
    const size_t in_rows = (size_t)nl + (size_t)kc.beta * kc.nt, job_rows = 2 * (size_t)kc.nt + 2 * (size_t)nl;
    const size_t budget = ctx->cfg.ksw_budget;                  // bytes of decomposition / accumulator scratch per group resp. chunk (SFG_KSW_BUDGET_MB)
    int in_grp = (int)(budget / (in_rows * N * 8)); if (in_grp < 1) in_grp = 1; if (in_grp > nin) in_grp = nin;
    int chunk = (int)(budget / (job_rows * N * 8)); if (chunk < 1) chunk = 1; if (chunk > nr) chunk = nr;
    const size_t ptr_bytes = (size_t)nr * (3 * sizeof(void *) + sizeof(int)) + 256;     // pointer tables for every job of a group
    // the forward NTTs run out of place (two half-row workgroups per row cannot share a buffer with their input): extT / ext2T receive the transforms
    const size_t ext_rows = (size_t)in_grp * kc.beta * kc.nt, ext2_rows = (size_t)chunk * 2 * nl;
    SFG_TRY(sfg_ws_reserve(ctx, (in_rows * in_grp + job_rows * chunk + ext_rows + ext2_rows) * N * 8 + ptr_bytes));
    u64 *c2 = (u64 *)ctx->ws, *ext = c2 + (size_t)in_grp * nl * N;
    u64 *acc = ext + ext_rows * N, *ext2 = acc + (size_t)chunk * 2 * kc.nt * N;
    u64 *extT = ext2 + ext2_rows * N, *ext2T = extT + ext_rows * N;
    const u64 **keys_all = (const u64 **)(ext2T + ext2_rows * N);
    const uint16_t **idx_all = (const uint16_t **)(keys_all + nr);
    u64 **out_all = (u64 **)(idx_all + nr);
    int *inidx_all = (int *)(out_all + nr);
    ModPattern pq; pq.period = nl; for (int m = 0; m < nl; m++) pq.m[m] = (int8_t)m;
    ModPattern pext; pext.period = kc.beta * kc.nt;
    for (int i = 0; i < kc.beta; i++) for (int t = 0; t < kc.nt; t++) pext.m[i * kc.nt + t] = kc.digit_of[t] == i ? (int8_t)-2 : (int8_t)kc.tmod[t];      // in-digit rows: nobody reads them
    ModPattern pp; pp.period = kc.np; for (int p = 0; p < kc.np; p++) pp.m[p] = (int8_t)(ctx->nq + p);
    for (int i0 = 0; i0 < nin; i0 += in_grp) {
        const int ni = nin - i0 < in_grp ? nin - i0 : in_grp;
        std::vector<int> jobs;
        for (int k = 0; k < nr; k++) if (job_in[k] >= i0 && job_in[k] < i0 + ni) jobs.push_back(k);
        if (jobs.empty()) continue;
        const u64 *bin = in + (size_t)i0 * ctw;
        // 1. c2 = INTT(c1) for every input of the group
        RowMap rm1; rm1.rpg = nl; rm1.gstride_in = ctw; rm1.gstride_out = (size_t)nl * N;
        SFG_TRY(launch_ntt_inv_map(ctx, bin + (size_t)nl * N, c2, (size_t)ni * nl, pq, rm1));
        // 2. digit extension + NTT (in-digit rows are skipped by the pattern)
        hipLaunchKernelGGL(k_ksw_extend, dim3(N / 256, kc.beta, ni), dim3(256), 0, ctx->stream, c2, bin, ext, kcd, ctx->modc);
        SFG_HIP(ctx, hipGetLastError());
        SFG_TRY(launch_ntt_fwd(ctx, ext, extT, (size_t)ni * kc.beta * kc.nt, pext));          // in-digit rows (pattern -1) are copied through
        // pointer tables of the whole group: staged in the pinned ring, stream-ordered (no host wait on the launch path)
        {
            const size_t nj = jobs.size();
            std::vector<const u64 *> kp(nj); std::vector<const uint16_t *> ip(nj); std::vector<u64 *> op(nj); std::vector<int> ii(nj);
            for (size_t k = 0; k < nj; k++) { int jb = jobs[k]; kp[k] = keyp[jb]; ip[k] = idxp[jb]; op[k] = outp[jb]; ii[k] = job_in[jb] - i0; }
            SFG_TRY(sfg_upload_small(ctx, keys_all, kp.data(), nj * sizeof(void *)));
            SFG_TRY(sfg_upload_small(ctx, idx_all, ip.data(), nj * sizeof(void *)));
            SFG_TRY(sfg_upload_small(ctx, out_all, op.data(), nj * sizeof(void *)));
            SFG_TRY(sfg_upload_small(ctx, inidx_all, ii.data(), nj * sizeof(int)));
        }
        for (size_t c0 = 0; c0 < jobs.size(); c0 += chunk) {
            const int nb = (int)(jobs.size() - c0 < (size_t)chunk ? jobs.size() - c0 : (size_t)chunk);
            const u64 **keys_d = keys_all + c0; const uint16_t **idx_d = idx_all + c0; u64 **out_d = out_all + c0; int *inidx_d = inidx_all + c0;
            // 3. inner product with the key
            hipLaunchKernelGGL(k_ksw_inner, dim3(N / 512, kc.nt, nb), dim3(256), 0, ctx->stream, extT, bin, keys_d, inidx_d, acc, kcd, ctx->modc, ctx->nmod);
            SFG_HIP(ctx, hipGetLastError());
            // 4. ModDown: INTT special rows in place, extend to Q, NTT
            RowMap rm4; rm4.rpg = kc.np; rm4.gstride_in = (size_t)kc.nt * N; rm4.gstride_out = (size_t)kc.nt * N;
            SFG_TRY(launch_ntt_inv_map(ctx, acc + (size_t)nl * N, acc + (size_t)nl * N, (size_t)nb * 2 * kc.np, pp, rm4));
            hipLaunchKernelGGL(k_moddown_extend, dim3(N / 256, 2, nb), dim3(256), 0, ctx->stream, acc, ext2, kcd, ctx->modc);
            SFG_HIP(ctx, hipGetLastError());
            SFG_TRY(launch_ntt_fwd(ctx, ext2, ext2T, (size_t)nb * 2 * nl, pq));
            // 5. finish + automorphism
            hipLaunchKernelGGL(k_ksw_finish, dim3(N / 256, nl, nb), dim3(256), 0, ctx->stream, bin, inidx_d, acc, ext2T, idx_d, out_d, kcd, ctx->modc,
                               add1 ? add1 + (size_t)i0 * nl * N : nullptr, ff);
            SFG_HIP(ctx, hipGetLastError());
        }
    }
    return 0;
}
// Rotate a batch.  `in` holds nin ciphertexts [nin][2][nl][N]; output j = RotateRight(in[in_index[j]], nrot[j])
// (RotateRightWithEvaluator semantics) written to out + j*ct words.  in_index == nullptr means identity (nin == nct).
int launch_rotate_right_indexed(sfg_ctx *ctx, const u64 *in, int nin, u64 *out, int nct, int level, const int *nrot_host, const int *in_index) {
    const int N = SFG_N, nl = level + 1;
    if (level < 0 || level >= ctx->nq) SFG_FAIL(ctx, "rotate: level out of range");
    const size_t ctw = (size_t)2 * nl * N;
    std::vector<int> job_in; std::vector<const u64 *> keyp; std::vector<const uint16_t *> idxp; std::vector<u64 *> outp;
    for (int j = 0; j < nct; j++) {
        int nrot = nrot_host[j] % SFG_SLOTS; if (nrot < 0) nrot += SFG_SLOTS;
        const int src = in_index ? in_index[j] : j;
        if (src < 0 || src >= nin) SFG_FAIL(ctx, "rotate: input index out of range");
        if (nrot == 0) {                                    // not rotated: copied; runs of consecutive (source, slot) pairs in one copy
            int run = 1;
            while (j + run < nct && (nrot_host[j + run] % SFG_SLOTS) == 0 && (in_index ? in_index[j + run] : j + run) == src + run && src + run < nin) run++;
            SFG_HIP(ctx, hipMemcpyAsync(out + j * ctw, in + (size_t)src * ctw, (size_t)run * ctw * 8, hipMemcpyDeviceToDevice, ctx->stream));
            j += run - 1;
            continue;
        }
        u64 g = sfg_galois_for_rotation(ctx, SFG_SLOTS - nrot);                 // basics.go:205: RotateNew(ct, slots - nrot)
        auto it = ctx->rotkeys().find(g);
        if (it == ctx->rotkeys().end()) SFG_FAIL(ctx, "rotate: no rotation key loaded for right-rotation by %d (galois element %llu)", nrot, g);
        job_in.push_back(src); keyp.push_back(it->second.key_dev); idxp.push_back(it->second.index_dev); outp.push_back(out + j * ctw);
    }
    return launch_keyswitch_jobs(ctx, in, nin, level, job_in, keyp, idxp, outp, nullptr);
}
// The same batch written as the MAC's fp64 operand rows: job j lands at outf + j * 2 * rowf (row pair of polynomials 0, 1), rowf = nplanes * N doubles.
// Jobs that do not rotate (nrot == 0) are converted from their input.
// out_slot (nullable): job j lands at outf + out_slot[j] * 2 * rowf instead of outf + j * 2 * rowf (sharded rotation-cache builds write job-major staging).
int launch_rotate_right_indexed_f64(sfg_ctx *ctx, const u64 *in, int nin, double *outf, int nct, int level, const int *nrot_host, const int *in_index, int L,
                                    const size_t *out_slot) {
    const int N = SFG_N, nl = level + 1;
    ctx->i8_gen++;                  // fp64 rot operand rows are (re)written: the int8 MAC's transposed copy of whatever buffer this is goes stale
    if (level < 0 || level >= ctx->nq || L > nl) SFG_FAIL(ctx, "rotate: level out of range");
    const size_t ctw = (size_t)2 * nl * N;
    std::vector<int> plane_of, is_big; const int nplanes = mac_dma_planes(ctx, L, plane_of, is_big);
    if (nplanes < 0) return 1;
    F64Form ff; memset(&ff, 0, sizeof ff);
    ff.on = 1; ff.L = L; ff.centre = mac_dma_packed_mask(ctx, L) != 0; ff.rowf = (size_t)nplanes * N;
    for (int l = 0; l < L; l++) { ff.plane_of[l] = plane_of[l]; ff.big[l] = is_big[l]; }
    std::vector<int> job_in; std::vector<const u64 *> keyp; std::vector<const uint16_t *> idxp; std::vector<u64 *> outp;
    for (int j = 0; j < nct; j++) {
        int nrot = nrot_host[j] % SFG_SLOTS; if (nrot < 0) nrot += SFG_SLOTS;
        const int src = in_index ? in_index[j] : j;
        if (src < 0 || src >= nin) SFG_FAIL(ctx, "rotate: input index out of range");
        double *dst = outf + (out_slot ? out_slot[j] : (size_t)j) * 2 * ff.rowf;
        if (nrot == 0) {                                    // not rotated: converted from the input; runs of consecutive (source, slot) pairs in one launch
            int run = 1;
            while (!out_slot && j + run < nct && (nrot_host[j + run] % SFG_SLOTS) == 0 && (in_index ? in_index[j + run] : j + run) == src + run && src + run < nin) run++;
            SFG_TRY(launch_rot_to_f64(ctx, in + (size_t)src * ctw, (size_t)2 * run, nl, L, dst));
            j += run - 1;
            continue;
        }
        u64 g = sfg_galois_for_rotation(ctx, SFG_SLOTS - nrot);
        auto it = ctx->rotkeys().find(g);
        if (it == ctx->rotkeys().end()) SFG_FAIL(ctx, "rotate: no rotation key loaded for right-rotation by %d (galois element %llu)", nrot, g);
        job_in.push_back(src); keyp.push_back(it->second.key_dev); idxp.push_back(it->second.index_dev); outp.push_back(reinterpret_cast<u64 *>(dst));
    }
    return launch_keyswitch_jobs(ctx, in, nin, level, job_in, keyp, idxp, outp, nullptr, &ff);
}
// relinearisation: out[i] = (tmp[i].p0 + d0, mid[i] + d1) with (d0, d1) = key switch of tmp[i].p1 under the key stored at Galois element 1
int launch_relinearize(sfg_ctx *ctx, const u64 *tmp, int nct, int level, const u64 *mid, u64 *out) {
    auto it = ctx->rotkeys().find(1);
    if (it == ctx->rotkeys().end()) SFG_FAIL(ctx, "relinearize: no relinearisation key loaded");
    const size_t ctw = (size_t)2 * (level + 1) * SFG_N;
    std::vector<int> job_in(nct); std::vector<const u64 *> keyp(nct, it->second.key_dev); std::vector<const uint16_t *> idxp(nct, it->second.index_dev); std::vector<u64 *> outp(nct);
    for (int j = 0; j < nct; j++) { job_in[j] = j; outp[j] = out + (size_t)j * ctw; }
    return launch_keyswitch_jobs(ctx, tmp, nct, level, job_in, keyp, idxp, outp, mid);
}
int launch_rotate_right(sfg_ctx *ctx, const u64 *in, u64 *out, int nct, int level, const int *nrot_host) {
    return launch_rotate_right_indexed(ctx, in, nct, out, nct, level, nrot_host, nullptr);
}

extern "C" int sfg_rotate_right_dev(sfg_ctx *ctx, const uint64_t *in, uint64_t *out, int nct, int level, const int *nrot_host) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (in == out) SFG_FAIL(ctx, "rotate: in and out must not alias");
    PhaseTimer t(ctx, "rotate");
    int rc = launch_rotate_right(ctx, (const u64 *)in, (u64 *)out, nct, level, nrot_host);
    t.stop(1);
    return rc;
}

// eval.ConjugateNew (crypto.ComplexConjugate / CReal, basics.go:826-846) and any other automorphism X -> X^g whose switching key was loaded under g
// (lattigo: g = 2N-1 conjugates the slots)
extern "C" int sfg_ct_galois_dev(sfg_ctx *ctx, const uint64_t *in, uint64_t *out, int nct, int level, uint64_t galois_el) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (in == out) SFG_FAIL(ctx, "galois: in and out must not alias");
    if (level < 0 || level >= ctx->nq) SFG_FAIL(ctx, "galois: level out of range");
    auto it = ctx->rotkeys().find(galois_el);
    if (it == ctx->rotkeys().end() || galois_el == 1) SFG_FAIL(ctx, "galois: no switching key loaded for galois element %llu", (unsigned long long)galois_el);
    const size_t ctw = (size_t)2 * (level + 1) * SFG_N;
    std::vector<int> job_in(nct); std::vector<const u64 *> keyp(nct, it->second.key_dev); std::vector<const uint16_t *> idxp(nct, it->second.index_dev); std::vector<u64 *> outp(nct);
    for (int j = 0; j < nct; j++) { job_in[j] = j; outp[j] = (u64 *)out + (size_t)j * ctw; }
    PhaseTimer t(ctx, "rotate");
    int rc = launch_keyswitch_jobs(ctx, (const u64 *)in, nct, level, job_in, keyp, idxp, outp, nullptr);
    t.stop(1);
    return rc;
}

int launch_ct_add(sfg_ctx *ctx, const u64 *a, const u64 *b, u64 *out, size_t nct, int level) {
    const int nl = level + 1; const size_t rows = nct * 2 * nl;
    if (!rows) return 0;
    hipLaunchKernelGGL(k_ct_add, dim3((unsigned)(rows * (SFG_N / 256))), dim3(256), 0, ctx->stream, a, b, out, nl, ctx->modc, rows);
    SFG_HIP(ctx, hipGetLastError());
    return 0;
}
extern "C" int sfg_ct_add_dev(sfg_ctx *ctx, const uint64_t *a, const uint64_t *b, uint64_t *out, int nct, int level) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (level < 0 || level >= ctx->nq) SFG_FAIL(ctx, "evaluator op: level %d out of range", level);
    if (nct < 0) SFG_FAIL(ctx, "evaluator op: negative ciphertext count");
    return launch_ct_add(ctx, (const u64 *)a, (const u64 *)b, (u64 *)out, (size_t)nct, level);
}
