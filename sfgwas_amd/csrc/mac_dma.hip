// mac_dma.hip — LDS-DMA form of the lazy-MAC batched modular GEMM (see mac.hip for the algebra and the
// reference lines it replaces: gwas/matmult.go:247-399).
//
// What changed relative to mac.hip's register-staged kernel: both operands now reach the CU through
// `global_load_lds_dwordx4` (async global -> LDS, no VGPR destination) into a ring of chunk slots, so the only
// per-thread state is the accumulator tile.  A chunk = 4 k-steps of (32 rot rows + 24 pt columns) x 16
// coefficients; up to 3 chunks are in flight per workgroup behind a counted `s_waitcnt vmcnt(N)` and ONE raw
// `s_barrier` per chunk (a __syncthreads() would drain the DMA queue).  The rotation cache is handed over already
// converted to fp64 (small moduli: one double per word; the 46-bit modulus: {low 23 bits, high bits} pairs), so
// a staged word is an FMA operand without any per-use conversion.
//
// Workgroup = 512 threads = 8 waves: wave = (row group rh < 4, column wave wc < 2); lane = (column group cg < 4,
// coefficient cc < 16); thread tile = 8 rows x 3 columns x 3 fp64 limb accumulators (144 VGPRs).
#include "common.hpp"
#include "kernels.hpp"
#include <algorithm>

constexpr int DM_CL = 16, DM_CT = 3, DM_CG = 4, DM_RG = 4, DM_RH = 8;
constexpr int DM_ROWS = DM_RG * DM_RH;                 // 32 rows per pass
constexpr int DM_KC = 4;

struct DmaArgs {
    const double *rotf;          // fp64 rotation cache, see k_rot_to_f64
    const u64 *pt; u64 *out;
    const u64 *zeros;            // >= 128 B of zeros: plaintext source of the padded k-steps of the last chunk
    size_t rotf_k_stride, rotf_r_stride;     // doubles
    size_t pt_k_stride, pt_n_stride;         // words
    size_t out_n_stride, out_r_stride;       // words
    int K, R, Ncols, L, accumulate, r0, l0, nl, flush, ntile;
    int plane0;                  // fp64 plane index of modulus l0 inside a rotf row
    int pt_half;                 // pt rows hold N/2 words: P[N-1-c] = P[c] (plaintexts of real slot vectors)
    size_t pt_l_stride;          // words between consecutive modulus rows of one plaintext (N or N/2)
};

// Packed-limb plaintext words (small moduli, q < 2^36) are the panel format of the DPP-broadcast kernel (mac_bc.hip; pack_limbs in common.hpp): this
// kernel, kept as the SFG_MAC_IMPL=dma baseline, reads plain canonical words and converts limbs with shift / mask / v_cvt_f64_u32.

// one 16-byte-per-lane LDS-DMA; lds_base must be wave-uniform (it goes to M0)
__device__ __forceinline__ void dma16(const void *gsrc, void *lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc, (__attribute__((address_space(3))) void *)lds_base, 16, 0, 0);
}

// WC = column waves per workgroup.  WC = 2: one 8-wave workgroup per CU (24 columns).  WC = 1: 4-wave workgroups of 12 columns,
// TWO per CU with a ring each: their barriers and DMA waits are not synchronised, so one computes while the other waits.
#ifdef SFG_AB          // the 8 x 3-tile LDS-DMA kernel of round 1: A/B build only (make ab)
template <bool BIG, int WC_> struct MacRing {
    // slot bytes: WC 2: 32 / 48 KiB, WC 1: 24 / 40 KiB
    static constexpr int RW = BIG ? 2 : 1, JOBS = (DM_KC * DM_ROWS * DM_CL * 8 * RW + DM_KC * DM_CG * DM_CT * WC_ * DM_CL * 8) / 1024;
    static constexpr int NW = 4 * WC_, A = (JOBS + NW - 1) / NW, SLOT = A * NW * 1024;
    static constexpr int DEPTH = WC_ == 2 ? (BIG ? 3 : 4) : (BIG ? 2 : 3);
    static constexpr int LDS = DEPTH * SLOT;
};
template <int WC_> struct MacGeom {
    static constexpr int WAVES = 4 * WC_, THREADS = 64 * WAVES, COLS = DM_CG * DM_CT * WC_;
};
template <bool BIG, int WC_>
__global__ void __launch_bounds__(64 * 4 * WC_, 2) k_mac_dma(DmaArgs a, const ModConst *modc) {
    constexpr int DM_COLS = MacGeom<WC_>::COLS, NWAVE = MacGeom<WC_>::WAVES;      // shadow the file-scope 2-wave-column geometry
    constexpr int RW = BIG ? 2 : 1;                                  // doubles per rot word
    constexpr int R_BYTES = DM_KC * DM_ROWS * DM_CL * 8 * RW;        // 16 KiB / 32 KiB
    constexpr int P_BYTES = DM_KC * DM_COLS * DM_CL * 8;             // 12 KiB (6 KiB for WC = 1)
    constexpr int R_JOBS = R_BYTES / 1024, P_JOBS = P_BYTES / 1024;  // 1 KiB per wave-instruction
    constexpr int JOBS = R_JOBS + P_JOBS;                            // 28 / 44 (22 / 38)
    constexpr int A = (JOBS + NWAVE - 1) / NWAVE;                    // DMA instructions per wave per chunk
    constexpr int SLOT = A * NWAVE * 1024;                           // slot incl. dummy jobs
    constexpr int DEPTH = MacRing<BIG, WC_>::DEPTH;                  // ring slots
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int N = SFG_N, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cc = lane & 15, cg = lane >> 4, rh = wave & 3, wc = wave >> 2;                  // wc == 0 when WC_ == 1
    // Block decode.  Column tiles that share one (c-block, modulus) slab of `rot` get consecutive slots on the same
    // XCD (blocks b and b+8 share an XCD).  With half-row plaintexts a c-block and its mirror (1023 - cblk) read
    // the same plaintext bytes, so the pair is placed back to back on one XCD and the second read is an L2 hit.
    int li, c0, tile; bool mirrored = false;
    {
        const int b = blockIdx.x;
        if (!a.pt_half) {
            const int grp = b / (8 * a.ntile), rem = b % (8 * a.ntile);
            const int slab = grp * 8 + (rem & 7); tile = rem >> 3;
            if (slab >= (N / DM_CL) * a.nl) return;
            li = slab / (N / DM_CL); c0 = (slab % (N / DM_CL)) * DM_CL;
        } else {
            const int per = 16 * a.ntile, grp = b / per, rem = b % per, idx = rem >> 3;
            const int sup = grp * 8 + (rem & 7); tile = idx % a.ntile; mirrored = idx >= a.ntile;
            if (sup >= (N / DM_CL / 2) * a.nl) return;
            li = sup / (N / DM_CL / 2);
            const int sb = sup % (N / DM_CL / 2);
            c0 = (mirrored ? (N / DM_CL - 1 - sb) : sb) * DM_CL;
        }
    }
    const int l = a.l0 + li;
    const double q = modc[l].q, qinv = modc[l].qinv;
    const int n0 = tile * DM_COLS + (wc * DM_CG + cg) * DM_CT;
    const int pcc = mirrored ? DM_CL - 1 - cc : cc;                     // P[N-1-c] = P[c]
    const int nchunk = (a.K + DM_KC - 1) / DM_KC;

    // ---- DMA source addressing.  Job j of a chunk moves 8 (k, row|col) pairs x 128 B; lane = (pair & 7, 16-B piece).
    // A source address is a wave-uniform base (SGPR pair: operand base of this workgroup, advanced per chunk on the scalar unit)
    // plus a per-lane 32-bit byte offset that never changes (VGPR): the saddr form of global_load_lds, so the loop carries no 64-bit
    // vector pointer arithmetic.  launch_mac_dma checks that the offsets fit 32 bits.
    const int pair_in_job = lane >> 3, piece = lane & 7;
    const int cp0 = mirrored ? N - DM_CL - c0 : c0;                    // mirror block start inside the half row
    const unsigned char *rot_u = (const unsigned char *)(a.rotf + (size_t)(a.plane0 + li * RW) * N + (BIG ? (size_t)c0 * 2 : (size_t)c0));
    const unsigned char *pt_u = (const unsigned char *)(a.pt + (size_t)tile * DM_COLS * a.pt_n_stride + (size_t)l * a.pt_l_stride + cp0);
    const unsigned char *z_u = (const unsigned char *)a.zeros;
    const size_t rot_step = (size_t)DM_KC * a.rotf_k_stride * 8, pt_step = (size_t)DM_KC * a.pt_k_stride * 8;
    // Job j = t * NWAVE + wave lands at slot + j * 1024: issue round t of every wave is a rot round (t < RT) or a pt round, known at
    // compile time.  R_JOBS is a multiple of NWAVE; pt rounds past P_JOBS re-load the last pt job into the slot's spare space.
    static_assert(R_JOBS % NWAVE == 0, "rot jobs must fill whole issue rounds");
    constexpr int RT = R_JOBS / NWAVE;
    unsigned off[A]; size_t poff[A]; int kk_of[A];                     // rot rounds: 32-bit lane offsets (host-checked); pt rounds: 64-bit
#pragma unroll
    for (int t = 0; t < A; t++) {
        const int job = t * NWAVE + wave;
        off[t] = 0; poff[t] = 0;
        if (t < RT) {
            int pr, cpart = 0;
            if (BIG) { pr = job * 4 + (pair_in_job >> 1); cpart = pair_in_job & 1; }   // 256 B per (k,row): 2 halves of 8 coefficients
            else pr = job * 8 + pair_in_job;
            const int kk = pr / DM_ROWS, r = pr % DM_ROWS;
            const int row = a.r0 + r < a.R ? a.r0 + r : a.R - 1;
            off[t] = (unsigned)((size_t)kk * a.rotf_k_stride * 8 + (size_t)row * a.rotf_r_stride * 8) + (BIG ? (unsigned)(cpart * 8 + piece) * 16u : (unsigned)piece * 16u);
            kk_of[t] = kk;
        } else {
            int pj = job - R_JOBS; pj = pj < P_JOBS ? pj : P_JOBS - 1;
            const int pr = pj * 8 + pair_in_job;
            const int kk = pr / DM_COLS, col = pr % DM_COLS;
            int n = tile * DM_COLS + col; n = n < a.Ncols ? n : a.Ncols - 1;
            poff[t] = (size_t)kk * a.pt_k_stride * 8 + (size_t)(n - tile * DM_COLS) * a.pt_n_stride * 8 + (size_t)piece * 16;
            kk_of[t] = kk;
        }
    }
    const unsigned zoff = (unsigned)piece * 16u;                      // zero plaintext words
    const int nchunk_full = a.K / DM_KC;                              // chunks whose 4 k-steps all exist
    // In the ragged last chunk (K % 4 != 0) the padded k-steps take a zero plaintext; the rot operand is read as is - the caller
    // guarantees that the (up to 3) k-slices after the last hold finite doubles (launch_mac_dma contract).
    auto issue_chunk = [&](int ch) {
        unsigned char *slot = lds + (size_t)(ch % DEPTH) * SLOT;
        const unsigned char *rb = rot_u + (size_t)ch * rot_step, *pb = pt_u + (size_t)ch * pt_step;
        if (ch < nchunk_full) {
#pragma unroll
            for (int t = 0; t < A; t++) dma16(t < RT ? rb + off[t] : pb + poff[t], slot + (t * NWAVE + wave) * 1024);
        } else {
#pragma unroll
            for (int t = 0; t < A; t++) {
                const unsigned char *src = t < RT ? rb + off[t] : pb + poff[t];
                if (t >= RT && ch * DM_KC + kk_of[t] >= a.K) src = z_u + zoff;
                dma16(src, slot + (t * NWAVE + wave) * 1024);
            }
        }
    };

    double acc[DM_RH][DM_CT][3];
#pragma unroll
    for (int r = 0; r < DM_RH; r++)
#pragma unroll
        for (int t = 0; t < DM_CT; t++) acc[r][t][0] = acc[r][t][1] = acc[r][t][2] = 0.0;

    // prologue: DEPTH-1 chunks in flight
#pragma unroll
    for (int ch = 0; ch < DEPTH - 1; ch++) if (ch < nchunk) issue_chunk(ch);

    int since_flush = 0;
#pragma unroll 1
    for (int ch = 0; ch < nchunk; ch++) {
        // chunk ch has landed once at most `ahead` younger chunks of this wave are still outstanding
        const int ahead = (nchunk - 1 - ch) < (DEPTH - 2) ? (nchunk - 1 - ch) : (DEPTH - 2);
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * A) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                 // everyone's pieces of chunk ch are in LDS; everyone is done with chunk ch-1
        if (ch + DEPTH - 1 < nchunk) issue_chunk(ch + DEPTH - 1);      // refill the slot chunk ch-1 just vacated
        const unsigned char *slot = lds + (size_t)(ch % DEPTH) * SLOT;
        const double *rbase = reinterpret_cast<const double *>(slot);
        const u64 *pbase = reinterpret_cast<const u64 *>(slot + R_BYTES);
        // software pipeline: the LDS words of k-step kk+1 are requested before the FMAs of k-step kk run
        double rcur[DM_RH * RW], rnxt[DM_RH * RW]; u64 pcur[DM_CT], pnxt[DM_CT];
        auto fetch = [&](int kk, double (&rr)[DM_RH * RW], u64 (&pp)[DM_CT]) {
#pragma unroll
            for (int t = 0; t < DM_CT; t++) pp[t] = pbase[(size_t)(kk * DM_COLS + (wc * DM_CG + cg) * DM_CT + t) * DM_CL + pcc];
#pragma unroll
            for (int r = 0; r < DM_RH; r++) {
                const int row = rh * DM_RH + r;
                if (BIG) { const double2 v2 = *reinterpret_cast<const double2 *>(rbase + ((size_t)(kk * DM_ROWS + row) * DM_CL + cc) * 2); rr[2 * r] = v2.x; rr[2 * r + 1] = v2.y; }
                else rr[r] = rbase[(size_t)(kk * DM_ROWS + row) * DM_CL + cc];
            }
        };
        auto fmas = [&](const double (&rr)[DM_RH * RW], const u64 (&pp)[DM_CT]) {
            double p0[DM_CT], p1[DM_CT], p2[DM_CT];
#pragma unroll
            for (int t = 0; t < DM_CT; t++) {
                const u64 p = pp[t];
                if (BIG) { p0[t] = (double)(unsigned)(p & 0x7FFFFFu); p1[t] = (double)(unsigned)(p >> 23); p2[t] = p0[t] + p1[t]; }   // Karatsuba: p2 = p_lo + p_hi
                else {
                    const unsigned plo = (unsigned)p, phi = (unsigned)(p >> 32);
                    p0[t] = (double)(plo & 0xFFFu); p1[t] = (double)((plo >> 12) & 0xFFFu); p2[t] = (double)((plo >> 24) | (phi << 8));
                }
            }
#pragma unroll
            for (int r = 0; r < DM_RH; r++) {
#pragma unroll
                for (int t = 0; t < DM_CT; t++) {
                    if (BIG) {
                        // 3 products per MAC: lo*lo, hi*hi and (lo+hi)*(lo+hi); the middle limb is recovered in the epilogue
                        acc[r][t][0] = __builtin_fma(rr[2 * r], p0[t], acc[r][t][0]);
                        acc[r][t][2] = __builtin_fma(rr[2 * r + 1], p1[t], acc[r][t][2]);
                        acc[r][t][1] = __builtin_fma(rr[2 * r] + rr[2 * r + 1], p2[t], acc[r][t][1]);
                    } else {
                        acc[r][t][0] = __builtin_fma(rr[r], p0[t], acc[r][t][0]);
                        acc[r][t][1] = __builtin_fma(rr[r], p1[t], acc[r][t][1]);
                        acc[r][t][2] = __builtin_fma(rr[r], p2[t], acc[r][t][2]);
                    }
                }
            }
        };
        fetch(0, rcur, pcur);
        fetch(1, rnxt, pnxt); fmas(rcur, pcur);
        fetch(2, rcur, pcur); fmas(rnxt, pnxt);
        fetch(3, rnxt, pnxt); fmas(rcur, pcur);
        fmas(rnxt, pnxt);
        static_assert(DM_KC == 4, "the pipeline above is written for 4 k-steps per chunk");
        since_flush += DM_KC;
        if (since_flush >= a.flush) {
            since_flush = 0;
#pragma unroll
            for (int r = 0; r < DM_RH; r++)
#pragma unroll
                for (int t = 0; t < DM_CT; t++) {
                    acc[r][t][0] = pred(acc[r][t][0], q, qinv); acc[r][t][1] = pred(acc[r][t][1], q, qinv); acc[r][t][2] = pred(acc[r][t][2], q, qinv);
                }
        }
    }
    constexpr double S1 = BIG ? 8388608.0 : 4096.0;
    const double s1 = S1, s1q = S1 / q;
    const double s2 = canon(S1 * S1, q, qinv), s2q = s2 / q;
    // epilogue: all previous-value loads are issued first (one wait), then the tile is reduced and stored;
    // a load->add->store chain per element would serialize 24 HBM round trips per thread
    u64 oldv[DM_RH][DM_CT];
#pragma unroll
    for (int t = 0; t < DM_CT; t++)
#pragma unroll
        for (int r = 0; r < DM_RH; r++) {
            // unconditional loads from clamped (always valid) addresses: a branch per element would put a
            // vmcnt(0) behind every load (24 serialized HBM round trips per thread)
            const int n = n0 + t < a.Ncols ? n0 + t : a.Ncols - 1;
            const int row = a.r0 + rh * DM_RH + r < a.R ? a.r0 + rh * DM_RH + r : a.R - 1;
            oldv[r][t] = a.out[(size_t)n * a.out_n_stride + (size_t)row * a.out_r_stride + (size_t)l * N + c0 + cc];
        }
#pragma unroll
    for (int t = 0; t < DM_CT; t++) {
        const int n = n0 + t;
#pragma unroll
        for (int r = 0; r < DM_RH; r++) {
            const int row = a.r0 + rh * DM_RH + r;
            double x = pred(acc[r][t][0], q, qinv);
            const double mid = BIG ? pred(acc[r][t][1], q, qinv) - pred(acc[r][t][0], q, qinv) - pred(acc[r][t][2], q, qinv) : pred(acc[r][t][1], q, qinv);
            x += mulmod_lazy(mid, s1, s1q, q);
            x += mulmod_lazy(pred(acc[r][t][2], q, qinv), s2, s2q, q);
            x += a.accumulate ? u64_to_f64(oldv[r][t] & 0x000FFFFFFFFFFFFFULL) : 0.0;
            if (n < a.Ncols && row < a.R)
                a.out[(size_t)n * a.out_n_stride + (size_t)row * a.out_r_stride + (size_t)l * N + c0 + cc] = f64_to_u64(canon(x, q, qinv));
        }
    }
}
#endif

// rotation cache -> fp64 operand form.  in: [nct][2][nl][N] u64 ciphertext rows; out row (ct, poly) holds the
// planes of moduli 0..L-1: a "big" modulus (>= 2^36) takes 2N doubles {low 23 bits, high bits} interleaved per
// coefficient, a small one N doubles.
// centre != 0: small-modulus words are stored as the centred representative in (-q/2, q/2] (packed-limb panels)
__global__ void __launch_bounds__(256) k_rot_to_f64(const u64 *in, double *out, int nl, int L, size_t out_row_stride, const int *plane_of, const int *is_big,
                                                    int centre, const ModConst *modc) {
    const int N = SFG_N; const size_t rowl = blockIdx.x / (N / 256);         // over [ct*2][L]
    const size_t ctp = rowl / L; const int l = (int)(rowl % L);
    const size_t x = (blockIdx.x % (N / 256)) * 256 + threadIdx.x;
    const u64 w = in[(ctp * nl + l) * N + x];
    double *o = out + ctp * out_row_stride + (size_t)plane_of[l] * N;
    if (is_big[l]) {            // centred, then a SIGNED 23-bit split: |lo| <= 2^22, |hi| <= (q / 2 >> 23) + 1 - the Karatsuba middle term (lo + hi)(p_lo + p_hi) halves
        const u64 q = modc[l].qi;
        const long long wc = w > (q >> 1) ? (long long)w - (long long)q : (long long)w;
        const long long lo = ((wc + 4194304) & 0x7FFFFF) - 4194304, hi = (wc - lo) >> 23;
        o[2 * x] = (double)lo; o[2 * x + 1] = (double)hi;
    }
    else {
        const u64 q = modc[l].qi;
        o[x] = (centre && w > (q >> 1)) ? -u64_to_f64(q - w) : u64_to_f64(w);
    }
}
// plain canonical words -> packed-limb words for the small-modulus rows of a plaintext array [nrows = (k, n, l)][words]
__global__ void __launch_bounds__(256) k_pack_pt(const u64 *in, u64 *out, size_t words_per_row, int L, unsigned packed_mask) {
    const size_t row = blockIdx.y, x = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (x >= words_per_row) return;
    const u64 w = in[row * words_per_row + x];
    out[row * words_per_row + x] = ((packed_mask >> (row % L)) & 1u) ? pack_limbs(w) : w;
}

int mac_dma_set_attrs(sfg_ctx *ctx) {
#ifndef SFG_AB
    (void)ctx; return 0;
#else
    constexpr int lds_b2 = MacRing<true, 2>::LDS, lds_s2 = MacRing<false, 2>::LDS, lds_s1 = MacRing<false, 1>::LDS;
    auto kb2 = k_mac_dma<true, 2>; auto ks2 = k_mac_dma<false, 2>; auto ks1 = k_mac_dma<false, 1>;
    SFG_HIP(ctx, hipFuncSetAttribute((const void *)ks2, hipFuncAttributeMaxDynamicSharedMemorySize, lds_s2));
    SFG_HIP(ctx, hipFuncSetAttribute((const void *)kb2, hipFuncAttributeMaxDynamicSharedMemorySize, lds_b2));
    SFG_HIP(ctx, hipFuncSetAttribute((const void *)ks1, hipFuncAttributeMaxDynamicSharedMemorySize, lds_s1));
    return 0;
#endif
}
// bit l set: the plaintext rows of modulus l use the packed-limb format (small moduli with the default broadcast kernel; not with the A/B build's plain panel / dma / reg kernels)
unsigned mac_dma_packed_mask(sfg_ctx *ctx, int L) {
    if (ctx->cfg.mac_plain_pt || ctx->cfg.mac_reg || !ctx->cfg.mac_bc) return 0u;       // the packed format is the broadcast kernel's
    unsigned m = 0; for (int l = 0; l < L; l++) if (ctx->q[l] < (1ULL << 36)) m |= 1u << l;
    return m;
}

// 46/47-bit moduli: rot words are centred and split into signed halves (k_rot_to_f64: |lo| <= 2^22, |hi| <= (q >> 24) + 1), plaintext words into
// unsigned halves (p_lo < 2^23, p_hi <= q >> 23).  The largest product of a k-step is the Karatsuba middle term.
double mac_big_maxterm(u64 q) { return (4194304.0 + (double)((q >> 24) + 1)) * (8388608.0 + (double)((q >> 23) + 1)); }
int mac_dma_planes(sfg_ctx *ctx, int L, std::vector<int> &plane_of, std::vector<int> &is_big) {
    plane_of.assign(L, 0); is_big.assign(L, 0); int nplanes = 0;
    for (int l = 0; l < L; l++) {
        if (ctx->q[l] >= (1ULL << 47)) { ctx->err = "sfg_mac: modulus >= 2^47 unsupported by the fp64 limb schedule"; return -1; }
        is_big[l] = ctx->q[l] >= (1ULL << 36); plane_of[l] = nplanes; nplanes += is_big[l] ? 2 : 1;
    }
    return nplanes;
}

// convert nrows polynomial rows [nrows][nl_rot][N] (ciphertexts are two consecutive rows) to the fp64 operand form rotf[nrows][nplanes*N]
int launch_rot_to_f64(sfg_ctx *ctx, const u64 *rot, size_t nrows, int nl_rot, int L, double *rotf) {
    ctx->i8_gen++;                  // (as launch_rotate_right_indexed_f64)
    const int centre = mac_dma_packed_mask(ctx, L) != 0;
    const int N = SFG_N;
    std::vector<int> plane_of, is_big; const int nplanes = mac_dma_planes(ctx, L, plane_of, is_big);
    if (nplanes < 0) return 1;
    // plane tables live in the context scratch pool (one small blocking upload per call would drain the stream)
    int *d_tab = nullptr; char name[32]; snprintf(name, sizeof name, "mac.planes.%d", L);
    const bool fresh = ctx->pool.find(name) == ctx->pool.end();
    SFG_TRY(sfg_scratch(ctx, name, 2 * L * sizeof(int), (void **)&d_tab));
    if (fresh) {
        SFG_HIP(ctx, hipMemcpy(d_tab, plane_of.data(), L * sizeof(int), hipMemcpyHostToDevice));
        SFG_HIP(ctx, hipMemcpy(d_tab + L, is_big.data(), L * sizeof(int), hipMemcpyHostToDevice));
    }
    hipLaunchKernelGGL(k_rot_to_f64, dim3((unsigned)(nrows * L * (N / 256))), dim3(256), 0, ctx->stream, rot, rotf, nl_rot, L, (size_t)nplanes * N, d_tab, d_tab + L, centre, ctx->modc);
    SFG_HIP(ctx, hipGetLastError());
    return 0;
}

// (kept for the call sites: the bias-free packed limbs of round 2 need no rot sums any more)
int launch_rot_sum(sfg_ctx *, const double *, size_t, int, int, double *) { return 0; }
int launch_pack_pt(sfg_ctx *ctx, const u64 *in, u64 *out, size_t nrows, size_t words_per_row, int L, unsigned packed_mask) {
    if (!nrows) return 0;
    if (nrows > 65535u * 1024u) SFG_FAIL(ctx, "pack_pt: too many rows");
    for (size_t r0 = 0; r0 < nrows; r0 += 65535u - 65535u % (unsigned)L) {          // grid.y bands that start on a multiple of L
        const size_t nr = std::min<size_t>(nrows - r0, 65535u - 65535u % (unsigned)L);
        hipLaunchKernelGGL(k_pack_pt, dim3((unsigned)((words_per_row + 255) / 256), (unsigned)nr), dim3(256), 0, ctx->stream, in + r0 * words_per_row, out + r0 * words_per_row,
                           words_per_row, L, packed_mask);
        SFG_HIP(ctx, hipGetLastError());
    }
    return 0;
}

// rotf: fp64 rotation cache with row stride nplanes*N doubles; rows_per_k = rows (ct, poly) between consecutive k.
// st.pt_packed: the small-modulus plaintext rows hold packed-limb words and rotf is centred (broadcast kernel only).
// Contract: when K % 4 != 0 the buffer must extend over the k-slices K .. 4*ceil(K/4)-1 and hold finite doubles there
// (they are multiplied by zero plaintexts).
int launch_mac_dma(sfg_ctx *ctx, const double *rotf, size_t rows_per_k, const u64 *pt, u64 *out, int K, int R, int Ncols, int L, int accumulate,
                   const MacStrides &st, const double *rotsum) {
    if (ctx->cfg.mac_bc && (st.pt_packed || mac_dma_packed_mask(ctx, L) == 0) && !ctx->cfg.mac_plain_pt)      // default: the DPP-broadcast kernel (mac_bc.hip)
        return launch_mac_bc(ctx, rotf, rows_per_k, pt, out, K, R, Ncols, L, accumulate, st, rotsum);
#ifndef SFG_AB
    SFG_FAIL(ctx, "the LDS-DMA baseline MAC kernel exists in the A/B build only (make ab)");
#else
    const int N = SFG_N;
    if (K <= 0 || R <= 0 || Ncols <= 0) return 0;
    if (!rotf) SFG_FAIL(ctx, "sfg_mac: internal: a rot operand given as int8 tiles only reached the fp64 kernel");
    std::vector<int> plane_of, is_big; const int nplanes = mac_dma_planes(ctx, L, plane_of, is_big);
    if (nplanes < 0) return 1;
    const size_t rowf = (size_t)nplanes * N;
    for (int r0 = 0; r0 < R; r0 += DM_ROWS) {
        int l = 0;
        while (l < L) {
            const bool big = is_big[l]; int e = l; while (e < L && is_big[e] == (int)big) e++;
            DmaArgs a; a.rotf = rotf; a.pt = pt; a.out = out; a.zeros = (const u64 *)ctx->zeros_dev();
            a.rotf_k_stride = rows_per_k * rowf; a.rotf_r_stride = rowf;
            a.pt_k_stride = st.pt_k; a.pt_n_stride = st.pt_n; a.out_n_stride = st.out_n; a.out_r_stride = st.out_r;
            a.K = K; a.R = R; a.Ncols = Ncols; a.L = L; a.accumulate = accumulate; a.r0 = r0; a.l0 = l; a.nl = e - l; a.plane0 = plane_of[l];
            a.pt_half = st.pt_half ? 1 : 0; a.pt_l_stride = st.pt_half ? N / 2 : N;
            if (st.pt_packed) SFG_FAIL(ctx, "sfg_mac: the LDS-DMA baseline kernel reads plain plaintext words");
            {   // the kernel addresses its operands as uniform base + 32-bit per-lane byte offset
                const double rot_max = (3.0 * (double)a.rotf_k_stride + (double)R * (double)a.rotf_r_stride) * 8.0 + 512.0;
                if (rot_max >= 4294967296.0) SFG_FAIL(ctx, "sfg_mac: operand strides exceed the 32-bit lane offsets of the DMA addressing (R = %d)", R);
            }
            // largest single term of a run: small moduli q * 2^12 (plain words, uncentred rot: this baseline kernel); big ones the Karatsuba middle
            // term (r_lo + r_hi) * (p_lo + p_hi), see mac_big_maxterm
            double maxterm = 0.0;
            for (int t = l; t < e; t++) if (big) { const double m = mac_big_maxterm(ctx->q[t]); if (m > maxterm) maxterm = m; }
            for (int t = l; t < e; t++) if (!big && (double)ctx->q[t] * 4096.0 > maxterm) maxterm = (double)ctx->q[t] * 4096.0;
            int f = (int)((9007199254740992.0 - 140737488355328.0) / maxterm); f = (f / DM_KC) * DM_KC;
            if (f < DM_KC) SFG_FAIL(ctx, "sfg_mac: flush period underflow");
            const int wcs = big ? 2 : ctx->cfg.mac_wc;       // column waves per workgroup for the small moduli
            const int cols_wg = DM_CG * DM_CT * wcs;
            a.flush = f; a.ntile = (Ncols + cols_wg - 1) / cols_wg;
            const int nslab = (st.pt_half ? N / DM_CL / 2 : N / DM_CL) * a.nl, ngrp = (nslab + 7) / 8;
            dim3 grid((unsigned)(ngrp * 8 * a.ntile * (st.pt_half ? 2 : 1)));
            PhaseTimer t(ctx, big ? "mac_big" : "mac_small");
            constexpr int lds_b2 = MacRing<true, 2>::LDS, lds_s2 = MacRing<false, 2>::LDS, lds_s1 = MacRing<false, 1>::LDS;
            auto kb2 = k_mac_dma<true, 2>; auto ks2 = k_mac_dma<false, 2>; auto ks1 = k_mac_dma<false, 1>;
                    if (big) hipLaunchKernelGGL(kb2, grid, dim3(512), lds_b2, ctx->stream, a, ctx->modc);
            else if (wcs == 2) hipLaunchKernelGGL(ks2, grid, dim3(512), lds_s2, ctx->stream, a, ctx->modc);
            else hipLaunchKernelGGL(ks1, grid, dim3(256), lds_s1, ctx->stream, a, ctx->modc);
            SFG_HIP(ctx, hipGetLastError());
            {   // algorithmic bytes of this launch: fp64 rot operand + plaintext words + accumulators written (and read when accumulating)
                const double nlm = (double)(e - l), rw = big ? 2.0 : 1.0, pw = st.pt_half ? 0.5 : 1.0;
                const double bytes = ((double)K * R * rw + (double)K * Ncols * pw + (double)Ncols * R * (accumulate ? 2.0 : 1.0)) * nlm * N * 8.0;
                t.stop(1, bytes);
            }
            l = e;
        }
    }
    return 0;
#endif
}
