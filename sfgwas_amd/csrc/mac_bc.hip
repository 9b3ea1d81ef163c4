// mac_bc.hip — DPP-broadcast form of the lazy-MAC batched modular GEMM (the default MAC kernel).
//
// Same algebra, operands and results as mac_dma.hip (gwas/matmult.go:247-399: MulCoeffsAndAdd128 / CPMultAccWithoutMRedV2 /
// ReduceAndAddUint128 net to the canonical sum  out[n][r][l][c] (+)= sum_k rot[k][r][l][c] * pt[k][n][l][c] mod q_l), and the same
// LDS-DMA ring (global_load_lds_dwordx4, counted vmcnt, one raw s_barrier per chunk of 4 k-steps).  What changed is the thread tile:
//
//   lane = (rho < 4 = DPP row = coefficient inside a quad, i < 16 = lane in the row = output column)
//   wave = (coefficient quad cq < 4, column group cgp < 2);  workgroup = 8 waves = 16 coefficients x 32 columns x ALL 30 rows
//   thread tile = 30 rows x 1 column x 3 fp64 limb accumulators = 90 accumulators
//
// Per coefficient the GEMM is an outer product rot[k][0..29] x pt[k][columns].  Lane i of a DPP row ALSO holds the rot words of rows i
// and 16 + i of that row's coefficient, and every FMA takes its rot operand from the lane that holds it through the fp64 ALU's own
// wavefront shuffle: `v_fmac_f64_dpp ... row_newbcast:r` (the one DPP control the DP ALU accepts on gfx90a+).  A k-step therefore
// costs a thread 2 rot reads + 1 plaintext read from LDS (24 B) and 3 v_perm_b32 for 90 FMAs, against 88 B and 9 for 72 FMAs in the
// 8 x 3 tile of mac_dma.hip - the LDS pipe (128 B/clk/CU, shared by the four SIMDs) drops from ~60 % to ~13 % of the FMA time, the
// 30 rows are not padded to 32, and the 46-bit modulus costs the same 90 FMAs (its lo + hi operand is one v_add_f64 per rot word).
//
// LDS images are laid out by the DMA's lane -> address map (any 16-byte granule of a 1 KiB job can come from anywhere), which is
// used to XOR-swizzle granules so that the 16 lanes of a DPP row (rows / columns i = 0..15 at one coefficient: a 128- or 256-byte
// stride in the natural layout) read 16 different bank groups.
#include "common.hpp"
#include "kernels.hpp"
#include <algorithm>
#include <cstdlib>
#include <utility>

constexpr int BC_CL = 16;          // coefficients per workgroup: 128-byte operand segments
constexpr int BC_KC = 4;           // k-steps per chunk
constexpr int BC_WAVES = 8;
constexpr int BC_COLS = 32;        // columns per workgroup
constexpr int BC_SROWS = 32;       // rot rows staged per k-step (rows past R are clamped duplicates)
constexpr int BC_MAXROWS = 30;     // rows per pass (s = 15 ciphertexts x 2 polynomials)

struct BcArgs {
    const double *rotf; const u64 *pt; u64 *out; const u64 *zeros;
    size_t rotf_k_stride, rotf_r_stride;     // doubles
    size_t pt_k_stride, pt_n_stride, pt_l_stride;   // words
    size_t out_n_stride, out_r_stride;       // words
    int K, R, Ncols, accumulate, r0, l0, nl, flush, ntile, plane0, pt_half;
    int diag;        // -DSFG_MAC_DIAG builds only (timing experiments, results invalid): 1 no rot DMA, 2 no pt DMA, 4 no barrier, 8 no FMAs
};

template <bool BIG> struct BcRing {
    static constexpr int RW = BIG ? 2 : 1;
    static constexpr int R_IMG = BC_SROWS * BC_CL * 8 * RW;        // rot image of one k-step: 4 / 8 KiB
    static constexpr int P_IMG = BC_COLS * BC_CL * 8;              // plaintext image of one k-step: 4 KiB
    static constexpr int R_BYTES = BC_KC * R_IMG, P_BYTES = BC_KC * P_IMG;
    static constexpr int SLOT = R_BYTES + P_BYTES;                 // 32 / 48 KiB
#ifndef BC_DEPTH_SMALL
#define BC_DEPTH_SMALL 5
#endif
    static constexpr int DEPTH = BIG ? 3 : BC_DEPTH_SMALL;         // 160 / 144 KiB of the 160 KiB
    static constexpr int LDS = DEPTH * SLOT;
    static constexpr int R_JOBS = R_BYTES / 1024, P_JOBS = P_BYTES / 1024;
    static constexpr int RT = R_JOBS / BC_WAVES, A = (R_JOBS + P_JOBS) / BC_WAVES;      // issue rounds per wave and chunk: rot rounds, all rounds
    static_assert(R_JOBS % BC_WAVES == 0 && P_JOBS % BC_WAVES == 0, "jobs must fill whole issue rounds");
};

// byte offset of the 8-byte word (row, coefficient c < 16) inside an image of 128-byte rows: 1 KiB jobs of 8 rows; inside a job the
// row order alternates with the job's parity and the 16-byte granules of row a are XORed with a
__device__ __forceinline__ int bc_swz8(int row, int c) {
    const int jb = row >> 3, a = row & 7;
    return jb * 1024 + ((a ^ (jb & 1)) * 128) + ((((c >> 1) ^ a)) * 16) + (c & 1) * 8;
}
// 16-byte {lo, hi} pairs of the 46-bit modulus: 256-byte rows, granule c XOR (row & 15)
__device__ __forceinline__ int bc_swz16(int row, int c) { return row * 256 + ((c ^ (row & 15)) * 16); }

__device__ __forceinline__ void bc_dma16(const void *gsrc, void *lds_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)gsrc, (__attribute__((address_space(3))) void *)lds_base, 16, 0, 0);
}
// acc += rot[lane LANE of this DPP row] * p
template <int LANE> __device__ __forceinline__ void fmac_bc(double &acc, double rot, double p) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(rot), "v"(p), "n"(LANE));      // volatile: the hand-made schedule (fetch / wait / FMA order) is kept as written
}
template <int LIMB, int... R> __device__ __forceinline__ void bc_rows(double (&acc)[sizeof...(R)][3], double ra, double rb, double p, std::integer_sequence<int, R...>) {
    (fmac_bc<(R & 15)>(acc[R][LIMB], R < 16 ? ra : rb, p), ...);
}

template <bool BIG, int ROWS>
__global__ void __launch_bounds__(64 * BC_WAVES, 2) k_mac_bc(BcArgs a, const ModConst *modc) {
    using Ring = BcRing<BIG>;
    constexpr int RT = Ring::RT, A = Ring::A, DEPTH = Ring::DEPTH, SLOT = Ring::SLOT, R_IMG = Ring::R_IMG, P_IMG = Ring::P_IMG, R_BYTES = Ring::R_BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int N = SFG_N, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rho = lane >> 4, i = lane & 15, cq = wave & 3, cgp = wave >> 2, cc = cq * 4 + rho;
    // block decode: as k_mac_dma - column tiles that share one (c-block, modulus) slab of `rot` sit on one XCD back to back, and so do
    // a c-block and its mirror, which read the same half-row plaintext bytes
    int li, c0, tile; bool mirrored = false;
    {
        const int b = blockIdx.x;
        if (!a.pt_half) {
            const int grp = b / (8 * a.ntile), rem = b % (8 * a.ntile);
            const int slab = grp * 8 + (rem & 7); tile = rem >> 3;
            if (slab >= (N / BC_CL) * a.nl) return;
            li = slab / (N / BC_CL); c0 = (slab % (N / BC_CL)) * BC_CL;
        } else {
            const int per = 16 * a.ntile, grp = b / per, rem = b % per, idx = rem >> 3;
            const int sup = grp * 8 + (rem & 7); tile = idx % a.ntile; mirrored = idx >= a.ntile;
            if (sup >= (N / BC_CL / 2) * a.nl) return;
            li = sup / (N / BC_CL / 2);
            const int sb = sup % (N / BC_CL / 2);
            c0 = (mirrored ? (N / BC_CL - 1 - sb) : sb) * BC_CL;
        }
    }
    const int l = a.l0 + li;
    const double q = modc[l].q, qinv = modc[l].qinv;
    const int pcc = mirrored ? BC_CL - 1 - cc : cc;                    // P[N-1-c] = P[c]
    const int nchunk = (a.K + BC_KC - 1) / BC_KC, nchunk_full = a.K / BC_KC;

    // ---- DMA source addressing: wave-uniform operand base (advanced per chunk on the scalar unit) + constant per-lane byte offsets
    const int cp0 = mirrored ? N - BC_CL - c0 : c0;
    const unsigned char *rot_u = (const unsigned char *)(a.rotf + (size_t)(a.plane0 + li * Ring::RW) * N + (BIG ? (size_t)c0 * 2 : (size_t)c0));
    const unsigned char *pt_u = (const unsigned char *)(a.pt + (size_t)tile * BC_COLS * a.pt_n_stride + (size_t)l * a.pt_l_stride + cp0);
    const unsigned char *z_u = (const unsigned char *)a.zeros + (BIG ? 0 : 256) + (lane & 7) * 16;     // zero plaintext words (packed zeros at +256 B)
    const size_t rot_step = (size_t)BC_KC * a.rotf_k_stride * 8, pt_step = (size_t)BC_KC * a.pt_k_stride * 8;
    unsigned roff[RT]; size_t poff[A - RT]; int pkk[A - RT];
#pragma unroll
    for (int t = 0; t < A; t++) {
        const int job = t * BC_WAVES + wave;
        if (t < RT) {
            int kk, row, g16;                                          // g16: 16-byte granule inside the row's segment
            if (BIG) { kk = job >> 3; row = (job & 7) * 4 + (lane >> 4); g16 = (lane & 15) ^ (row & 15); }
            else { const int jb = job & 3, av = (lane >> 3) ^ (jb & 1); kk = job >> 2; row = jb * 8 + av; g16 = (lane & 7) ^ av; }
            const int rr = a.r0 + row < a.R ? a.r0 + row : a.R - 1;
            roff[t] = (unsigned)((size_t)kk * a.rotf_k_stride * 8 + (size_t)rr * a.rotf_r_stride * 8) + (unsigned)g16 * 16u;
        } else {
            const int pj = job - Ring::R_JOBS, jb = pj & 3, av = (lane >> 3) ^ (jb & 1), kk = pj >> 2, col = jb * 8 + av;
            int n = tile * BC_COLS + col; n = n < a.Ncols ? n : a.Ncols - 1;
            poff[t - RT] = (size_t)kk * a.pt_k_stride * 8 + (size_t)(n - tile * BC_COLS) * a.pt_n_stride * 8 + (size_t)((lane & 7) ^ av) * 16;
            pkk[t - RT] = kk;
        }
    }
    // DMA issue rounds [t0, t1) of chunk ch (A rounds per wave and chunk; round t moves job t * 8 + wave).  The rounds of one chunk are issued
    // a few at a time between the k-steps of the chunk that runs meanwhile: 32 KiB landing in one burst would hold the LDS ports long
    // enough to delay the operand reads of the running k-step.
    auto issue_rounds = [&](int ch, int t0, int t1) {
        unsigned char *slot = lds + (size_t)(ch % DEPTH) * SLOT;
        const unsigned char *rb = rot_u + (size_t)ch * rot_step, *pb = pt_u + (size_t)ch * pt_step;
        const bool ragged = ch >= nchunk_full;     // ragged last chunk: the padded k-steps take a zero plaintext; rot is read as is (finite by the launcher's contract)
#pragma unroll
        for (int t = 0; t < A; t++) {
            if (t < t0 || t >= t1) continue;
            if (t < RT) {
#ifdef SFG_MAC_DIAG
                if (a.diag & 1) continue;
#endif
                bc_dma16(rb + roff[t], slot + (t * BC_WAVES + wave) * 1024);
            } else {
#ifdef SFG_MAC_DIAG
                if (a.diag & 2) continue;
#endif
                bc_dma16(ragged && ch * BC_KC + pkk[t - RT] >= a.K ? z_u : pb + poff[t - RT], slot + (t * BC_WAVES + wave) * 1024);
            }
        }
    };
    auto issue_chunk = [&](int ch) { issue_rounds(ch, 0, A); };
    // round groups of the four issue points of the loop
    constexpr int G1 = (A + 3) / 4, G2 = G1 + (A + 2) / 4, G3 = G2 + (A + 1) / 4;

    double acc[ROWS][3];
#pragma unroll
    for (int r = 0; r < ROWS; r++) acc[r][0] = acc[r][1] = acc[r][2] = 0.0;
    // every ring slot is filled before the loop starts
#pragma unroll
    for (int ch = 0; ch < DEPTH; ch++) if (ch < nchunk) issue_chunk(ch);

    // ---- LDS operand reads, issued and awaited by hand.  (Left to the compiler, every FMA group is preceded by s_waitcnt lgkmcnt(0),
    // which also waits for the prefetch of the NEXT k-step that was issued just before it.)  A fetch is NLDS ds_read instructions; the
    // wait that makes a fetch's registers valid names them as in/out operands, so no consumer can be scheduled ahead of it.
    typedef double d2v __attribute__((ext_vector_type(2)));
    struct Opd { d2v ra, rb; u64 p; };                                // small moduli: ra = {row i, row 16 + i}; 46-bit modulus: ra = {lo, hi} of row i, rb of row 16 + i
    constexpr int NLDS = BIG ? 3 : 2;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)lds;
    const unsigned r_thr = lds0 + (unsigned)(BIG ? bc_swz16(i, cc) : bc_swz8(i, cc));       // row 16 + i sits 2048 (BIG: 4096) bytes further
    const unsigned p_thr = lds0 + (unsigned)bc_swz8(cgp * 16 + i, pcc);
    unsigned r_cur = r_thr, p_cur = p_thr;                            // + byte offset of the current ring slot
#define BC_FETCH(KK, O) do { \
        if (BIG) { \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"((O).ra) : "v"(r_cur), "n"((KK) * R_IMG) : "memory"); \
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"((O).rb) : "v"(r_cur), "n"((KK) * R_IMG + 4096) : "memory"); \
        } else asm volatile("ds_read2st64_b64 %0, %1 offset0:%2 offset1:%3" : "=v"((O).ra) : "v"(r_cur), "n"((KK) * R_IMG / 512), "n"((KK) * R_IMG / 512 + 4) : "memory"); \
        asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"((O).p) : "v"(p_cur), "n"(R_BYTES + (KK) * P_IMG) : "memory"); \
    } while (0)
#define BC_WAIT(NOUT, O) do { \
        if (BIG) asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"((O).ra), "+v"((O).rb), "+v"((O).p) : "n"(NOUT) : "memory"); \
        else asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"((O).ra), "+v"((O).p) : "n"(NOUT) : "memory"); \
    } while (0)
    // small moduli: the (subnormal) limb doubles x_k * 2^-1034 live in registers; a k-step rewrites only their HIGH dwords (one v_perm_b32 each), and the
    // accumulators are exact multiples of 2^-1034: the periodic fold uses q and 1/q scaled accordingly (both normal numbers)
    double pl[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { int z; asm volatile("v_mov_b32 %0, 0" : "=v"(z)); pl[k] = __hiloint2double(0, z); }
    const double qf = BIG ? q : (q * 0x1p-517) * 0x1p-517, qfinv = BIG ? qinv : (qinv * 0x1p517) * 0x1p517;
    constexpr auto rows = std::make_integer_sequence<int, ROWS>{};
    auto fmas = [&](const Opd &o) {
#ifdef SFG_MAC_DIAG
        if (a.diag & 8) return;
#endif
        if (BIG) {      // Karatsuba: lo*lo, hi*hi, (lo+hi)*(lo+hi); the middle limb is recovered in the epilogue
            double p0 = (double)(unsigned)(o.p & 0x7FFFFFu), p1 = (double)(unsigned)(o.p >> 23), p2 = p0 + p1;
            double sa = o.ra.x + o.ra.y, sb = o.rb.x + o.rb.y;
            // a VGPR written by the VALU may not be read by a DPP instruction in the next two issue slots, and the hazard recognizer
            // does not look inside inline asm: everything the VALU just produced passes through this barrier first
            asm volatile("s_nop 1" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(sa), "+v"(sb));
            bc_rows<0>(acc, o.ra.x, o.rb.x, p0, rows); bc_rows<2>(acc, o.ra.y, o.rb.y, p1, rows); bc_rows<1>(acc, sa, sb, p2, rows);
        } else {
            const unsigned plo = (unsigned)o.p, phi = (unsigned)(o.p >> 32);
            pl[0] = __hiloint2double((int)__builtin_amdgcn_perm(plo, 0u, 0x0C05040Cu), __double2loint(pl[0]));      // x0 * 2^-1034: high dword 00 | x0 | 00
            pl[1] = __hiloint2double((int)__builtin_amdgcn_perm(plo, 0u, 0x0C07060Cu), __double2loint(pl[1]));      // x1 * 2^-1034
            pl[2] = __hiloint2double((int)__builtin_amdgcn_perm(phi, 0u, 0x0C05040Cu), __double2loint(pl[2]));      // x2 * 2^-1034
            asm volatile("s_nop 1" : "+v"(pl[0]), "+v"(pl[1]), "+v"(pl[2]));       // DPP hazard barrier, as above
            bc_rows<0>(acc, o.ra.x, o.ra.y, pl[0], rows); bc_rows<1>(acc, o.ra.x, o.ra.y, pl[1], rows); bc_rows<2>(acc, o.ra.x, o.ra.y, pl[2], rows);
        }
    };
    // chunk `next` has landed in every wave's view once this wave's own pieces have (counted vmcnt: chunks issued after it may still be in
    // flight) and the workgroup has passed the barrier; the barrier also tells that everybody has finished READING the current chunk
    auto sync_for = [&](int next) {
        int ahead = (nchunk - 1 < next + DEPTH - 2 ? nchunk - 1 : next + DEPTH - 2) - next;      // chunks issued after `next`
        if (DEPTH == 3 && ahead > 1) ahead = 1;
        if (ahead >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * A) : "memory");
        else if (ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * A) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef SFG_MAC_DIAG
        if (!(a.diag & 4))
#endif
        __builtin_amdgcn_s_barrier();
    };
    static_assert(DEPTH <= 5 && 3 * A < 64, "vmcnt budget");
    Opd oa, ob;
    unsigned slot_off = 0;
    sync_for(0);
    BC_FETCH(0, oa);
    int since_flush = 0;
#pragma unroll 1
    for (int ch = 0; ch < nchunk; ch++) {
        // the slot of chunk ch - 1 was handed back at the end of the previous iteration; its refill (chunk ch - 1 + DEPTH) goes out in four parts
        const int refill = ch - 1 + DEPTH; const bool do_refill = ch >= 1 && refill < nchunk;
        if (do_refill) issue_rounds(refill, G1, G2);
        BC_FETCH(1, ob); BC_WAIT(NLDS, oa); fmas(oa);
        if (do_refill) issue_rounds(refill, G2, G3);
        BC_FETCH(2, oa); BC_WAIT(NLDS, ob); fmas(ob);
        if (do_refill) issue_rounds(refill, G3, A);
        BC_FETCH(3, ob); BC_WAIT(NLDS, oa); fmas(oa);
        BC_WAIT(0, ob);                               // this wave has read everything it needs from chunk ch
        if (ch + 1 < nchunk) {
            // Before the last k-step's FMAs: hand over to chunk ch + 1 (its first operands travel from LDS while those 90 FMAs run, so the
            // barrier is not followed by an exposed LDS round trip) and start refilling the slot of chunk ch, which nobody reads any more.
            sync_for(ch + 1);
            if (ch + DEPTH < nchunk) issue_rounds(ch + DEPTH, 0, G1);
            slot_off = slot_off + SLOT == (unsigned)(DEPTH * SLOT) ? 0u : slot_off + SLOT;
            r_cur = r_thr + slot_off; p_cur = p_thr + slot_off;
            BC_FETCH(0, oa);
        }
        fmas(ob);
        static_assert(BC_KC == 4, "the pipeline above is written for 4 k-steps per chunk");
        since_flush += BC_KC;
        if (since_flush >= a.flush) {
            since_flush = 0;
#pragma unroll
            for (int r = 0; r < ROWS; r++) { acc[r][0] = pred(acc[r][0], qf, qfinv); acc[r][1] = pred(acc[r][1], qf, qfinv); acc[r][2] = pred(acc[r][2], qf, qfinv); }
        }
    }
#undef BC_FETCH
#undef BC_WAIT
    // ---- epilogue: limb recombination, canonical store
    constexpr double S1 = BIG ? 8388608.0 : 4096.0;
    const double s1 = S1, s1q = S1 / q;
    const double s2 = canon(S1 * S1, q, qinv), s2q = s2 / q;
    const int n = tile * BC_COLS + cgp * 16 + i;
    const int nclamp = n < a.Ncols ? n : a.Ncols - 1;
    u64 *ocol = a.out + (size_t)nclamp * a.out_n_stride + (size_t)l * N + c0 + cc;
    constexpr int HALF = (ROWS + 1) / 2;
#pragma unroll
    for (int h = 0; h < 2; h++) {
        // all previous-value loads of a half first (one wait), from clamped (always valid) addresses
        u64 oldv[HALF];
#pragma unroll
        for (int x = 0; x < HALF; x++) {
            const int r = h * HALF + x;
            const int row = a.r0 + r < a.R ? a.r0 + r : a.R - 1;
            oldv[x] = a.accumulate ? ocol[(size_t)row * a.out_r_stride] : 0ULL;
        }
#pragma unroll
        for (int x = 0; x < HALF; x++) {
            const int r = h * HALF + x;
            if (r >= ROWS) continue;
            const int row = a.r0 + r;
            // small moduli: the sums are multiples of 2^-1034 (< 2^53 of them): two exact scalings give the integers back
            const double i0 = BIG ? acc[r][0] : (acc[r][0] * 0x1p517) * 0x1p517, i1 = BIG ? acc[r][1] : (acc[r][1] * 0x1p517) * 0x1p517,
                         i2 = BIG ? acc[r][2] : (acc[r][2] * 0x1p517) * 0x1p517;
            const double a0 = pred(i0, q, qinv), a1 = pred(i1, q, qinv), a2 = pred(i2, q, qinv);
            double v = a0;
            const double mid = BIG ? a1 - a0 - a2 : a1;
            v += mulmod_lazy(mid, s1, s1q, q);
            v += mulmod_lazy(a2, s2, s2q, q);
            v += u64_to_f64(oldv[x] & 0x000FFFFFFFFFFFFFULL);
            if (n < a.Ncols && row < a.R) ocol[(size_t)row * a.out_r_stride] = f64_to_u64(canon(v, q, qinv));
        }
    }
}

template <bool BIG, int ROWS> static int bc_set_attr(sfg_ctx *ctx) {
    auto k = k_mac_bc<BIG, ROWS>;
    SFG_HIP(ctx, hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, BcRing<BIG>::LDS));
    return 0;
}
int mac_bc_set_attrs(sfg_ctx *ctx) {
    SFG_TRY((bc_set_attr<false, 30>(ctx))); SFG_TRY((bc_set_attr<false, 26>(ctx))); SFG_TRY((bc_set_attr<false, 16>(ctx))); SFG_TRY((bc_set_attr<false, 10>(ctx))); SFG_TRY((bc_set_attr<false, 4>(ctx)));
    SFG_TRY((bc_set_attr<true, 30>(ctx))); SFG_TRY((bc_set_attr<true, 26>(ctx))); SFG_TRY((bc_set_attr<true, 16>(ctx))); SFG_TRY((bc_set_attr<true, 10>(ctx))); SFG_TRY((bc_set_attr<true, 4>(ctx)));
    return 0;
}
template <bool BIG, int ROWS> static void bc_launch(sfg_ctx *ctx, dim3 grid, const BcArgs &a) {
    hipLaunchKernelGGL((k_mac_bc<BIG, ROWS>), grid, dim3(64 * BC_WAVES), BcRing<BIG>::LDS, ctx->stream, a, ctx->modc);
}

// Same contract as launch_mac_dma (mac_dma.hip).  The small-modulus plaintext rows must be in the packed-limb format.
int launch_mac_bc(sfg_ctx *ctx, const double *rotf, size_t rows_per_k, const u64 *pt, u64 *out, int K, int R, int Ncols, int L, int accumulate,
                  const MacStrides &st, const double *rotsum) {
    const int N = SFG_N;
    if (K <= 0 || R <= 0 || Ncols <= 0) return 0;
    std::vector<int> plane_of, is_big; const int nplanes = mac_dma_planes(ctx, L, plane_of, is_big);
    if (nplanes < 0) return 1;
    const size_t rowf = (size_t)nplanes * N;
    for (int r0 = 0; r0 < R; r0 += BC_MAXROWS) {
        const int rows = std::min(BC_MAXROWS, R - r0);
        int l = 0;
        while (l < L) {
            const bool big = is_big[l]; int e = l; while (e < L && is_big[e] == (int)big) e++;
            if (!big && !st.pt_packed) SFG_FAIL(ctx, "sfg_mac: the broadcast MAC needs packed-limb plaintext rows for the small moduli");
            // (the int8 kernels take <= 96 columns and K 2^14 ND < 2^31 per launch; anything else - a DiagCache product over more than 96 block columns - stays here)
            const bool i8_fits = Ncols <= 96 && (long long)K * 6 < 131072;
            if (!big && st.i8 && st.pt_half && (i8_fits || st.pt_digits)) {            // the 35-bit moduli on the int8 matrix core (mac_i8.hip)
                SFG_TRY(launch_mac_i8_small(ctx, rotf, rows_per_k * rowf, rowf, plane_of[l], pt, out, K, r0 + rows, r0, Ncols, l, e - l, accumulate, st));      // (row bound r0 + rows: this pass's rows only - the kernel would take 32 from r0)
                l = e; continue;
            }
            if (big && st.i8_big && st.pt_half && (i8_fits || st.pt_digits_big)) {         // the 46-bit modulus likewise, six digits (one modulus per launch)
                MacStrides sb = st; sb.pt_digits = st.pt_digits_big;
                for (int t = l; t < e; t++) SFG_TRY(launch_mac_i8_big(ctx, rotf, rows_per_k * rowf, rowf, plane_of[t], pt, out, K, r0 + rows, r0, Ncols, t, accumulate, sb));
                l = e; continue;
            }
            if (!rotf) SFG_FAIL(ctx, "sfg_mac: internal: a rot operand given as int8 tiles only reached the fp64 kernel");
            BcArgs a; a.rotf = rotf; a.pt = pt; a.out = out; a.zeros = (const u64 *)ctx->zeros_dev();
            a.rotf_k_stride = rows_per_k * rowf; a.rotf_r_stride = rowf;
            a.pt_k_stride = st.pt_k; a.pt_n_stride = st.pt_n; a.pt_l_stride = st.pt_half ? N / 2 : N; a.out_n_stride = st.out_n; a.out_r_stride = st.out_r;
            a.K = K; a.R = R; a.Ncols = Ncols; a.accumulate = accumulate; a.r0 = r0; a.l0 = l; a.nl = e - l; a.plane0 = plane_of[l]; a.pt_half = st.pt_half ? 1 : 0;
            {   // rot lane offsets are 32-bit
                const double rot_max = (3.0 * (double)a.rotf_k_stride + (double)R * (double)a.rotf_r_stride) * 8.0 + 512.0;
                if (rot_max >= 4294967296.0) SFG_FAIL(ctx, "sfg_mac: operand strides exceed the 32-bit lane offsets of the DMA addressing (R = %d)", R);
            }
            // largest single term of a run (in units of 2^-1034 for the small moduli): q / 2 * 2^12 (centred rot x 12-bit limb), big ones the Karatsuba middle term
            double maxterm = 0.0;
            for (int t = l; t < e; t++) {
                const double m = big ? mac_big_maxterm(ctx->q[t]) : (double)ctx->q[t] * 2048.0;
                if (m > maxterm) maxterm = m;
            }
            int f = (int)((9007199254740992.0 - 140737488355328.0) / maxterm); f = (f / BC_KC) * BC_KC;
            if (f < BC_KC) SFG_FAIL(ctx, "sfg_mac: flush period underflow");
            a.diag = 0;
#ifdef SFG_MAC_DIAG
            if (const char *e = getenv("SFG_MAC_DIAG")) a.diag = atoi(e);
#endif
            a.flush = f; a.ntile = (Ncols + BC_COLS - 1) / BC_COLS;
            const int nslab = (st.pt_half ? N / BC_CL / 2 : N / BC_CL) * a.nl, ngrp = (nslab + 7) / 8;
            dim3 grid((unsigned)(ngrp * 8 * a.ntile * (st.pt_half ? 2 : 1)));
            PhaseTimer t(ctx, big ? "mac_big" : "mac_small");
            // row-count instances: 30 (kp = 15, PCA), 26 (s = 13 = ncov + 1 + npc + 2: the association scan's concat, assoc.go:699-704), 16, 10 (s = ncov = 5, gWY), 4 (s <= 2)
            if (big) { if (rows > 26) bc_launch<true, 30>(ctx, grid, a); else if (rows > 16) bc_launch<true, 26>(ctx, grid, a); else if (rows > 10) bc_launch<true, 16>(ctx, grid, a);
                       else if (rows > 4) bc_launch<true, 10>(ctx, grid, a); else bc_launch<true, 4>(ctx, grid, a); }
            else { if (rows > 26) bc_launch<false, 30>(ctx, grid, a); else if (rows > 16) bc_launch<false, 26>(ctx, grid, a); else if (rows > 10) bc_launch<false, 16>(ctx, grid, a);
                   else if (rows > 4) bc_launch<false, 10>(ctx, grid, a); else bc_launch<false, 4>(ctx, grid, a); }
            SFG_HIP(ctx, hipGetLastError());
            {   // algorithmic bytes of this launch (as launch_mac_dma): fp64 rot operand + plaintext words + accumulators written (and read when accumulating)
                const double nlm = (double)(e - l), rw = big ? 2.0 : 1.0, pw = st.pt_half ? 0.5 : 1.0;
                const double bytes = ((double)K * rows * rw + (double)K * Ncols * pw + (double)Ncols * rows * (accumulate ? 2.0 : 1.0)) * nlm * N * 8.0;
                t.stop(1, bytes);
            }
            l = e;
        }
    }
    return 0;
}
