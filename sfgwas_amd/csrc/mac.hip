// mac.hip — the hot loop of the hot path: lazy multiply-accumulate of rotated ciphertext rows with encoded
// diagonals (MulCoeffsAndAdd128 / CPMultAccWithoutMRedV2 / ReduceAndAddUint128, gwas/matmult.go:247-399).
//
// The reference walks diagonals outermost and keeps s*d*m_ct u128 accumulator polynomials in RAM (440 GB at
// 100k x 1M).  Here the same sums are organised as a batched modular GEMM: for every coefficient c and modulus l
//       out[n][r][l][c] (+)= sum_{k<K} rot[k][r][l][c] * pt[k][n][l][c]   (mod q_l)
// k = (block row, baby step), n = (giant step, block column), r = (ciphertext row i, poly).  Coefficients are
// independent, so nothing is shared across lanes except the `rot` operand, which every column of a workgroup
// re-uses: it is staged once per workgroup in LDS (as fp64), each `pt` word is read exactly once from HBM
// straight into registers, and the R accumulators of a column never leave VGPRs until the K loop ends.
//
// Arithmetic (measured: v_fma_f64 and v_mad_u64_u32 issue at the same rate on gfx950, fp64 needs no carry
// chain): exact integer dot products in fp64 limbs —
//   q < 2^36 ("small"): pt = p0 + p1*2^12 + p2*2^24 (12-bit limbs), acc_j += rot * p_j      3 FMA / MAC
//   q < 2^47 ("big")  : rot = r0 + r1*2^23, pt = p0 + p1*2^23,  acc00,acc01,acc11            4 FMA / MAC
// every product is < 2^48 so >= 24 terms add exactly below 2^53; accumulators are folded to (-q,q) every
// `flush` terms, recombined and canonically reduced once at the end.  The canonical sum is what
// MForm + u128 MAC + REDC + eval.Reduce produce in the reference, so outputs are bit-identical.
#include "common.hpp"
#include "kernels.hpp"

constexpr int MAC_CL = 16;        // coefficients per workgroup (lanes 0..15 of each 16-lane group)
constexpr int MAC_CG = 4;         // column groups per wave
constexpr int MAC_CT = 3;         // columns per thread
constexpr int MAC_RGRP = 4;       // row groups (waves that share columns but own different rows)
constexpr int MAC_WC = 2;         // column waves
constexpr int MAC_THREADS = 64 * MAC_RGRP * MAC_WC;           // 512 threads = 8 waves = 2 per SIMD (256 VGPRs each)
constexpr int MAC_COLS = MAC_CG * MAC_CT * MAC_WC;            // 24 output columns per workgroup
constexpr int MAC_KC = 4;         // k-steps staged per LDS chunk
constexpr int MAC_RMAX = 32;      // rows per pass (2 * kp = 30 for kp = 15, pca.go:87); 8 per thread

struct MacArgs {
    const u64 *rot; const u64 *pt; u64 *out;
    size_t rot_k_stride, rot_r_stride;     // words between consecutive k / consecutive rows of `rot` (row holds >= L modulus rows)
    size_t pt_k_stride, pt_n_stride;       // words between consecutive k / consecutive columns of `pt`
    size_t out_n_stride, out_r_stride;     // words between consecutive columns / rows of `out`
    int K, R, Ncols, L, accumulate;
    int r0;            // first row of this pass
    int l0, nl;        // moduli handled by this launch [l0, l0+nl)
    int flush;         // fold accumulators every `flush` k-steps (multiple of MAC_KC)
    int ntile;
};

// RH = rows per thread (a quarter of the rows of a pass); thread tile = RH rows x 3 columns.
// LDS image of one chunk: [kk][c][row] with the 2*RH rows of a coefficient contiguous, so a thread pulls its
// rows with ds_read_b128 (two rows per read; 16 distinct 16-byte words per wave-read at a stride of 2*RH*8 B,
// which spreads over all 64 banks) and every value read feeds 2 columns x 3..4 FMAs.
#ifdef SFG_AB          // the register-staged kernel of round 1: A/B build only (make ab)
template <bool BIG, int RH>
__global__ void __launch_bounds__(MAC_THREADS, 2) k_mac(MacArgs a, const ModConst *modc) {
    constexpr int RW = BIG ? 2 : 1;                    // doubles per staged rot word
    constexpr int RT = MAC_RGRP * RH;                  // rows per pass
    constexpr int RP = (RT * RW + 1) & ~1;             // row-vector length in doubles, padded to 16 B
    __shared__ __attribute__((aligned(16))) double lds[2][MAC_KC][MAC_CL][RP];
    const int N = SFG_N, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int cc = lane & 15, cg = lane >> 4, rh = wave % MAC_RGRP, wc = wave / MAC_RGRP;
    // XCD-aware decode: the `ntile` column tiles that share one (c-block, modulus) slab of `rot` get consecutive
    // slots on the same XCD (blocks b and b+8 share an XCD), so the slab is served from that XCD's L2.
    const int b = blockIdx.x, grp = b / (8 * a.ntile), rem = b % (8 * a.ntile);
    const int slab = grp * 8 + (rem & 7), tile = rem >> 3;
    const int nslab = (N / MAC_CL) * a.nl;
    if (slab >= nslab) return;
    const int l = a.l0 + slab / (N / MAC_CL), c0 = (slab % (N / MAC_CL)) * MAC_CL;
    const double q = modc[l].q, qinv = modc[l].qinv;
    const int n0 = tile * MAC_COLS + (wc * MAC_CG + cg) * MAC_CT;     // this thread's first output column

    const size_t rot_k_stride = a.rot_k_stride, pt_k_stride = a.pt_k_stride;
    const u64 *rot_base = a.rot + (size_t)a.r0 * a.rot_r_stride + (size_t)l * N + c0;
    const u64 *pt_ptr[MAC_CT];          // walks k: advanced by pt_k_stride after every load
#pragma unroll
    for (int t = 0; t < MAC_CT; t++) { int nn = n0 + t < a.Ncols ? n0 + t : a.Ncols - 1; pt_ptr[t] = a.pt + (size_t)nn * a.pt_n_stride + (size_t)l * N + c0 + cc; }
    int k_loaded = 0;                    // next k to fetch

    double acc[RH][MAC_CT][3];
#pragma unroll
    for (int r = 0; r < RH; r++)
#pragma unroll
        for (int t = 0; t < MAC_CT; t++) acc[r][t][0] = acc[r][t][1] = acc[r][t][2] = 0.0;

    const int nchunk = (a.K + MAC_KC - 1) / MAC_KC;
    // stage chunk ch into buffer bufi: word e -> (kk, r, c); global reads are 128-B segments (16 c of one row)
    auto stage = [&](int ch, int bufi) {
        for (int e = tid; e < MAC_KC * RT * MAC_CL; e += MAC_THREADS) {
            int kk = e / (RT * MAC_CL), rm = e % (RT * MAC_CL), r = rm / MAC_CL, c = rm % MAC_CL;
            int k = ch * MAC_KC + kk;
            u64 w = 0;
            if (k < a.K && a.r0 + r < a.R) w = rot_base[(size_t)k * rot_k_stride + (size_t)r * a.rot_r_stride + c];
            if (BIG) {
                lds[bufi][kk][c][r * 2 + 0] = (double)(unsigned)(w & 0x7FFFFFu);
                lds[bufi][kk][c][r * 2 + 1] = u64_to_f64(w >> 23);
            } else {
                lds[bufi][kk][c][r] = u64_to_f64(w);
            }
        }
    };
    // plaintext words: two-deep register ring, refilled two k-steps ahead.  The k loop is deliberately NOT
    // unrolled beyond that ring: a fully unrolled chunk lets the scheduler hoist every LDS read of the chunk and
    // spill the accumulators.
    u64 pa[MAC_CT], pb[MAC_CT];
    auto load_p = [&](u64 (&p)[MAC_CT]) {
        const bool ok = k_loaded < a.K;
#pragma unroll
        for (int t = 0; t < MAC_CT; t++) { p[t] = ok ? *pt_ptr[t] : 0ULL; pt_ptr[t] += ok ? pt_k_stride : 0; }
        k_loaded++;
    };
    auto kstep = [&](int k, u64 (&p)[MAC_CT]) {
        const double *rv = &lds[(k / MAC_KC) & 1][k % MAC_KC][cc][rh * RH * RW];
        double p0[MAC_CT], p1[MAC_CT], p2[MAC_CT];
#pragma unroll
        for (int t = 0; t < MAC_CT; t++) {
            if (BIG) { p0[t] = (double)(unsigned)(p[t] & 0x7FFFFFu); p1[t] = u64_to_f64(p[t] >> 23); p2[t] = 0.0; }
            else {
                const unsigned plo = (unsigned)p[t], phi = (unsigned)(p[t] >> 32);
                p0[t] = (double)(plo & 0xFFFu); p1[t] = (double)((plo >> 12) & 0xFFFu); p2[t] = (double)((plo >> 24) | (phi << 8));
            }
        }
        load_p(p);
#pragma unroll
        for (int r = 0; r < RH; r++) {
#pragma unroll
            for (int t = 0; t < MAC_CT; t++) {
                if (BIG) {
                    const double r0 = rv[2 * r], r1 = rv[2 * r + 1];
                    acc[r][t][0] = __builtin_fma(r0, p0[t], acc[r][t][0]);
                    acc[r][t][1] = __builtin_fma(r0, p1[t], acc[r][t][1]);
                    acc[r][t][1] = __builtin_fma(r1, p0[t], acc[r][t][1]);
                    acc[r][t][2] = __builtin_fma(r1, p1[t], acc[r][t][2]);
                } else {
                    const double x = rv[r];
                    acc[r][t][0] = __builtin_fma(x, p0[t], acc[r][t][0]);
                    acc[r][t][1] = __builtin_fma(x, p1[t], acc[r][t][1]);
                    acc[r][t][2] = __builtin_fma(x, p2[t], acc[r][t][2]);
                }
            }
        }
    };
    auto flush_all = [&]() {
#pragma unroll
        for (int r = 0; r < RH; r++)
#pragma unroll
            for (int t = 0; t < MAC_CT; t++) {
                acc[r][t][0] = pred(acc[r][t][0], q, qinv); acc[r][t][1] = pred(acc[r][t][1], q, qinv); acc[r][t][2] = pred(acc[r][t][2], q, qinv);
            }
    };
    stage(0, 0);
    load_p(pa);
    load_p(pb);
    __syncthreads();
    int since_flush = 0;
#pragma unroll 1
    for (int ch = 0; ch < nchunk; ch++) {
        if (ch + 1 < nchunk) stage(ch + 1, (ch + 1) & 1);
        const int kbase = ch * MAC_KC;
#pragma unroll 1
        for (int kk = 0; kk < MAC_KC; kk += 2) {
            kstep(kbase + kk, pa);
            kstep(kbase + kk + 1, pb);
        }
        since_flush += MAC_KC;
        if (since_flush >= a.flush) { since_flush = 0; flush_all(); }
        __syncthreads();
    }
    // recombine limbs: value = acc0 + acc1 * 2^S + acc2 * 2^(2S)  (mod q)
    constexpr double S1 = BIG ? 8388608.0 : 4096.0;
    const double s1 = S1, s1q = S1 / q;
    const double s2 = canon(S1 * S1, q, qinv), s2q = s2 / q;     // 2^46 (big) may exceed q: reduce it first
#pragma unroll
    for (int t = 0; t < MAC_CT; t++) {
        const int n = n0 + t;
        if (n >= a.Ncols) continue;
#pragma unroll
        for (int r = 0; r < RH; r++) {
            const int row = a.r0 + rh * RH + r;
            if (row < a.R) {
                double x = pred(acc[r][t][0], q, qinv);
                x += mulmod_lazy(pred(acc[r][t][1], q, qinv), s1, s1q, q);
                x += mulmod_lazy(pred(acc[r][t][2], q, qinv), s2, s2q, q);
                u64 *o = a.out + (size_t)n * a.out_n_stride + (size_t)row * a.out_r_stride + (size_t)l * N + c0 + cc;
                if (a.accumulate) x += u64_to_f64(*o);
                *o = f64_to_u64(canon(x, q, qinv));
            }
        }
    }
}

template <bool BIG>
static int launch_mac_rt(sfg_ctx *ctx, MacArgs a, int rt) {
    const int nslab = (SFG_N / MAC_CL) * a.nl;
    a.ntile = (a.Ncols + MAC_COLS - 1) / MAC_COLS;
    const int ngrp = (nslab + 7) / 8;
    dim3 grid((unsigned)(ngrp * 8 * a.ntile));
    if (rt <= 4) hipLaunchKernelGGL((k_mac<BIG, 1>), grid, dim3(MAC_THREADS), 0, ctx->stream, a, ctx->modc);
    else if (rt <= 16) hipLaunchKernelGGL((k_mac<BIG, 4>), grid, dim3(MAC_THREADS), 0, ctx->stream, a, ctx->modc);
    else hipLaunchKernelGGL((k_mac<BIG, MAC_RMAX / MAC_RGRP>), grid, dim3(MAC_THREADS), 0, ctx->stream, a, ctx->modc);
    SFG_HIP(ctx, hipGetLastError());
    return 0;
}
#endif

int launch_mac(sfg_ctx *ctx, const u64 *rot, const u64 *pt, u64 *out, int K, int R, int Ncols, int L, int accumulate) {
    MacStrides st;
    st.rot_k = (size_t)R * L * SFG_N; st.rot_r = (size_t)L * SFG_N;
    st.pt_k = (size_t)Ncols * L * SFG_N; st.pt_n = (size_t)L * SFG_N;
    st.out_n = (size_t)R * L * SFG_N; st.out_r = (size_t)L * SFG_N;
    return launch_mac_strided(ctx, rot, pt, out, K, R, Ncols, L, accumulate, st);
}

int launch_mac_strided(sfg_ctx *ctx, const u64 *rot, const u64 *pt, u64 *out, int K, int R, int Ncols, int L, int accumulate, const MacStrides &st) {
#ifndef SFG_AB
    (void)rot; (void)pt; (void)out; (void)K; (void)R; (void)Ncols; (void)L; (void)accumulate; (void)st;
    SFG_FAIL(ctx, "the register-staged MAC kernel exists in the A/B build only (make ab)");
#else
    if (K <= 0 || R <= 0 || Ncols <= 0) return 0;
    if (L < 1 || L > ctx->nq) SFG_FAIL(ctx, "sfg_mac: L out of range");
    for (int r0 = 0; r0 < R; r0 += MAC_RMAX) {
        int rt = R - r0 < MAC_RMAX ? R - r0 : MAC_RMAX;
        // moduli are handled in runs of equal kind (small / big)
        int l = 0;
        while (l < L) {
            bool big = ctx->q[l] >= (1ULL << 36);
            if (ctx->q[l] >= (1ULL << 47)) SFG_FAIL(ctx, "sfg_mac: modulus >= 2^47 unsupported by the fp64 limb schedule");
            int e = l; while (e < L && (ctx->q[e] >= (1ULL << 36)) == big) e++;
            MacArgs a{rot, pt, out, st.rot_k, st.rot_r, st.pt_k, st.pt_n, st.out_n, st.out_r, K, R, Ncols, L, accumulate, r0, l, e - l, 0, 0};
            // largest exact run: terms are < 2^48 (two per MAC in the big acc01), partial sums must stay < 2^53
            double maxterm = big ? 2.0 * 16777216.0 * 16777216.0 : (double)ctx->q[l] * 4096.0;
            for (int t = l; t < e; t++) if (!big && (double)ctx->q[t] * 4096.0 > maxterm) maxterm = (double)ctx->q[t] * 4096.0;
            int f = (int)((9007199254740992.0 - 140737488355328.0) / maxterm);
            f = (f / MAC_KC) * MAC_KC; if (f < MAC_KC) SFG_FAIL(ctx, "sfg_mac: flush period underflow");
            a.flush = f;
            {
                PhaseTimer t(ctx, big ? "mac_big" : "mac_small");      // HIP events on the launch stream around this kernel
                int rc = big ? launch_mac_rt<true>(ctx, a, rt) : launch_mac_rt<false>(ctx, a, rt);
                t.stop(1);
                if (rc) return rc;
            }
            l = e;
        }
    }
    return 0;
#endif
}

// (A/B build: SFG_MAC_IMPL=reg selects the register-staged kernel of this file)
bool mac_use_dma(const sfg_ctx *ctx) { return !ctx->cfg.mac_reg; }

extern "C" int sfg_mac_dev(sfg_ctx *ctx, const uint64_t *rot, const uint64_t *pt, uint64_t *out, int K, int R, int Ncols, int L, int accumulate) {
    SFG_HIP(ctx, hipSetDevice(ctx->device));
    if (L < 1 || L > ctx->nq) SFG_FAIL(ctx, "sfg_mac: L out of range");
    if (K < 1 || R < 1 || Ncols < 1) SFG_FAIL(ctx, "sfg_mac: K, R and Ncols must be positive (got %d, %d, %d)", K, R, Ncols);
    PhaseTimer t(ctx, "mac");
    int rc;
    if (!mac_use_dma(ctx)) rc = launch_mac(ctx, (const u64 *)rot, (const u64 *)pt, (u64 *)out, K, R, Ncols, L, accumulate);
    else {
        std::vector<int> plane_of, is_big; const int nplanes = mac_dma_planes(ctx, L, plane_of, is_big);
        if (nplanes < 0) return 1;
        double *rotf = nullptr;
        const size_t rows = (size_t)K * R;                       // rot is [K][R][L][N]
        const size_t pad_rows = (size_t)((4 - K % 4) % 4) * R;       // k-slices read (against zero plaintexts) by the ragged last chunk
        SFG_HIP(ctx, hipMalloc(&rotf, (rows + pad_rows) * (size_t)nplanes * SFG_N * 8));
        if (pad_rows) SFG_HIP(ctx, hipMemsetAsync(rotf + rows * (size_t)nplanes * SFG_N, 0, pad_rows * (size_t)nplanes * SFG_N * 8, ctx->stream));
        rc = launch_rot_to_f64(ctx, (const u64 *)rot, rows, L, L, rotf);
        MacStrides st;
        st.rot_k = (size_t)R * L * SFG_N; st.rot_r = (size_t)L * SFG_N;
        st.pt_k = (size_t)Ncols * L * SFG_N; st.pt_n = (size_t)L * SFG_N;
        st.out_n = (size_t)R * L * SFG_N; st.out_r = (size_t)L * SFG_N;
        // default build: the small-modulus plaintext rows go through the packed-limb format the product path uses (A/B build, plain panel: as given)
        const unsigned pmask = mac_dma_packed_mask(ctx, L);
        u64 *ptp = nullptr; double *rsum = nullptr;
        if (!rc && pmask) {
            const size_t prows = (size_t)K * Ncols * L;
            if (hipMalloc(&ptp, prows * SFG_N * 8) != hipSuccess || hipMalloc(&rsum, (size_t)R * nplanes * SFG_N * 8) != hipSuccess) { rc = 1; ctx->err = "sfg_mac: out of device memory"; }
            if (!rc) rc = launch_pack_pt(ctx, (const u64 *)pt, ptp, prows, SFG_N, L, pmask);
            if (!rc) rc = launch_rot_sum(ctx, rotf, (size_t)R, K, L, rsum);
            st.pt_packed = true;
        }
        if (!rc) rc = launch_mac_dma(ctx, rotf, (size_t)R, pmask ? ptp : (const u64 *)pt, (u64 *)out, K, R, Ncols, L, accumulate, st, rsum);
        (void)hipStreamSynchronize(ctx->stream); (void)hipFree(rotf); (void)hipFree(ptp); (void)hipFree(rsum);
    }
    t.stop(1);
    return rc;
}
