// i8_move.hpp - the int8 MAC's plaintext transposition as a LOW-OCCUPANCY mover (round 6): the same [k][coefficient] -> [coefficient][16 k] byte transposition as
// k_i8_pack_pt_digits (mac_i8.hip), written to run as ONE workgroup per CU beside the plaintext NTT's workgroups instead of five per CU on its own.
//
// Why: the pass is HBM bound (6.3 TB/s with the chip to itself) and the plaintext NTT is fp64-issue bound (0.3 of the HBM rate); one after the other they add up.
// k_i8_pack_pt_digits hides its memory latency behind OTHER workgroups of its kind (32 KiB of loads in flight per workgroup, two barriers per digit), so the few that
// fit beside NTT workgroups are latency bound (profiles/r05_fused_ntt_pack_ubench.txt).  Here a workgroup walks a list of units (unit = one digit plane of one
// (modulus, column tile, 16 k, 128 coefficients) item: 256 rows of 128 bytes in, 128 pieces of 256 bytes out) with the loads of the next DEPTH - 1 units in flight in
// registers while the current one goes through the LDS image: (DEPTH - 1) x 32 KiB per CU always outstanding, no other workgroup needed.  Same LDS image, same stores,
// same bytes as the pass (tests/test_gpu_mover.py compares the tile buffers word by word).  VALU work is 8 v_perm per 16 bytes - nothing beside the NTT's butterflies.
#pragma once
#include "common.hpp"

struct I8Args {
    const double *rotf; const u64 *pt; u64 *out;
    size_t rotf_k_stride, rotf_r_stride, pt_k_stride, pt_n_stride, pt_l_stride, out_n_stride, out_r_stride;
    size_t pt_d_stride, pt_cb_stride;      // bytes between the digit planes of a modulus row / between its 128-byte coefficient blocks (N/2 and 128 unless the panel is K-major)
    size_t pt_l0_off;                      // words from a plaintext's start to the row of modulus l0; row of modulus l0 + m is pt_l_stride * m further (compact digit-plane panels: 5 or 6 planes apart)
    int K, R, Ncols, accumulate, r0, l0, nl, plane0, nch, njt, pt_digits;
    int kb;                                // 0: k is the row of the rot operand; else k' = g * kb + baby with baby < 91 real (streamed plaintext tiles: block rows start on a dword)
    int8_t *A, *B; u64 *T;
    int fake;                              // A/B build, microbenchmark only (results INVALID): bit 0 = the mover reads 32 KiB contiguous per unit, bit 1 = writes 32 KiB contiguous, bit 2 = sleeps instead of moving
};
constexpr int I8_PD = 128;                 // coefficients per transposition item (128-byte source runs)

// A slice of one MAC launch's plaintext transposition, carried by one kernel launch: items [first, first + count) of the n5 + n6 items of the panel
// (item < n5: five digit planes of a 35-bit modulus, args a5; else six planes of the 46-bit modulus, args a6), dealt round-robin to `nblocks` mover workgroups.
struct MoveJob {
    I8Args a5, a6;
    unsigned n5 = 0, n6 = 0, first = 0, count = 0, nblocks = 0;
    int depth = 3, nt = 0;                 // host side: which instance of the kernel (units in flight per workgroup; streaming loads and stores)
};

// One delayed MAC launch's transposition riding in the NTT launches of the next launch's encode (kernels.hpp).  The product rides with `nblocks` = cfg.pt_ride mover
// workgroups per NTT launch, one unit in flight, streaming loads and stores - the best point of profiles/r06_mover_ubench.txt.
struct PtRide {
    bool on = false;
    MoveJob job;                           // first / count are set per NTT launch
    unsigned next = 0, per = 0;            // next item to hand out; items per NTT launch
    unsigned total() const { return job.n5 + job.n6; }
    double item_bytes5 = 0, item_bytes6 = 0;      // bytes read + written per item (phase statistics)
};

typedef unsigned v4u __attribute__((ext_vector_type(4)));
struct I8MoveItem { const unsigned char *src; int8_t *dst; int kq, jt; };
template <int ND>
__device__ __forceinline__ I8MoveItem i8_move_item(const I8Args &a, unsigned item) {
    const int H = SFG_N / 2;
    unsigned b = item;
    const int cb = (int)(b % (H / I8_PD)); b /= H / I8_PD;
    const int kq = (int)(b % (unsigned)(a.nch * 4)); b /= (unsigned)(a.nch * 4);
    const int jt = (int)(b % (unsigned)a.njt), m = (int)(b / (unsigned)a.njt);
    I8MoveItem r;
    r.src = reinterpret_cast<const unsigned char *>(a.pt + a.pt_l0_off + (size_t)m * a.pt_l_stride) + (size_t)cb * a.pt_cb_stride;
    r.dst = a.B + (((((size_t)m * H + cb * I8_PD) * a.njt + jt) * a.nch + (kq >> 2)) * ND) * 1024 + (kq & 3) * 256;
    r.kq = kq; r.jt = jt;
    return r;
}
// Lane roles.  A unit is 256 rows (j < 16 columns, k < 16) of 128 bytes; a lane loads 16 bytes (cq8 = lane & 7: coefficients 16 cq8 .. + 15) of the four rows
// k = 4 k4 + x, x < 4 (k4 = lane >> 3 & 3) of column j = 8 it + 2 wave + (lane >> 5), it < 2: eight 16-byte loads per unit and thread, every address a SCALAR base
// (item, digit, it, wave, x) plus ONE per-thread 32-bit offset - no address registers to keep beside the 3 x 32 data registers in flight.  Rows that do not exist
// (k >= K, column >= Ncols) are read from the lane's k4 = 0 / even-column row instead (always inside the panel) and zeroed when the unit is finished.
struct I8MoveLane { unsigned voff, koff, joff; int cq8, k4, jb, wave; };
__device__ __forceinline__ I8MoveLane i8_move_lane(const I8Args &a, int tid) {
    I8MoveLane l;
    const int lane = tid & 63;
    l.wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    l.cq8 = lane & 7; l.k4 = (lane >> 3) & 3; l.jb = lane >> 5;
    l.koff = (unsigned)l.k4 * 4u * (unsigned)(a.pt_k_stride * 8); l.joff = (unsigned)l.jb * (unsigned)(a.pt_n_stride * 8);
    l.voff = (unsigned)l.cq8 * 16u + l.koff + l.joff;
    return l;
}
template <int ND, bool NT>
__device__ __forceinline__ void i8_move_issue(const I8Args &a, const I8MoveLane &l, unsigned item, int d, v4u (&w)[8]) {
    const I8MoveItem t = i8_move_item<ND>(a, item);
    const unsigned char *src = t.src + (size_t)d * a.pt_d_stride;
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int n0 = t.jt * 16 + it * 8 + l.wave * 2;                    // scalar: the even column of this wave's pair
        const unsigned jo = n0 + l.jb < a.Ncols ? 0u : l.joff;
#pragma unroll
        for (int x = 0; x < 4; x++) {
            const int k0 = t.kq * 16 + x;                                 // scalar: the lane's row is k0 + 4 k4
            // (no branch around a load: the compiler's vmcnt bookkeeping stays exact only in straight-line code, and exact counts are what keeps the later units in flight)
            const bool ok = n0 < a.Ncols && k0 < a.K;                     // scalar; else the item's first row, zeroed in i8_move_finish
            const unsigned vo = ok ? l.voff - jo - (k0 + 4 * l.k4 < a.K ? 0u : l.koff) : (unsigned)l.cq8 * 16u;
            const size_t so = ok ? ((size_t)n0 * a.pt_n_stride + (size_t)k0 * a.pt_k_stride) * 8 : ((size_t)(t.jt * 16) * a.pt_n_stride + (size_t)(t.kq * 16) * a.pt_k_stride) * 8;
            const v4u *p = reinterpret_cast<const v4u *>(src + so + vo);
#ifdef SFG_AB
            if (a.fake & 1) p = reinterpret_cast<const v4u *>(reinterpret_cast<const unsigned char *>(a.pt) + ((size_t)item * ND + d) * 32768 + (size_t)((it * 4 + x) * 256 + l.wave * 64 + (int)(threadIdx.x & 63)) * 16);
            if (a.fake & 8) {      // the read pattern of a panel laid out [column][plane][128-byte coefficient block][k][128 B]: the 16 k of a column are one 2 KiB run
                const int m = (int)(item / ((unsigned)(a.nch * 4) * (unsigned)(SFG_N / 2 / I8_PD)) / (unsigned)a.njt), cb = (int)(item % (SFG_N / 2 / I8_PD));
                const size_t pl = (ND == 6 ? 0 : 6 + m * 5) + d, n = (size_t)t.jt * 16 + it * 8 + l.wave * 2 + l.jb, k = (size_t)t.kq * 16 + 4 * l.k4 + x;
                p = reinterpret_cast<const v4u *>(reinterpret_cast<const unsigned char *>(a.pt) + (((n * 26 + pl) * 64 + cb) * ((size_t)a.nch * 64) + k) * 128 + l.cq8 * 16);
            }
#endif
            w[it * 4 + x] = NT ? __builtin_nontemporal_load(p) : *p;
        }
    }
}
// through the image [c 128][j 16][k4 4] dwords (j XORed with bits 4..6 of c: the 32 lanes of a store hit 32 banks, the 16-byte reads are conflict free as they are)
// into the item's 128 pieces of 256 bytes
template <int ND, bool NT>
__device__ __forceinline__ void i8_move_finish(const I8Args &a, const I8MoveLane &l, unsigned item, int d, v4u (&w)[8], unsigned *img, int tid) {
    const I8MoveItem t = i8_move_item<ND>(a, item);
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int j = it * 8 + l.wave * 2 + l.jb;
        const bool jv = t.jt * 16 + j < a.Ncols;
        unsigned r[4][4];
#pragma unroll
        for (int x = 0; x < 4; x++) {
            const bool v = jv && t.kq * 16 + 4 * l.k4 + x < a.K;
            const v4u q = w[it * 4 + x];
            r[x][0] = v ? q.x : 0u; r[x][1] = v ? q.y : 0u; r[x][2] = v ? q.z : 0u; r[x][3] = v ? q.w : 0u;
        }
        unsigned *o = img + (l.cq8 * 16) * 64 + ((j ^ l.cq8) << 2) + l.k4;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            unsigned t4[4]; bytes_tr4(r[0][e], r[1][e], r[2][e], r[3][e], t4);
#pragma unroll
            for (int i = 0; i < 4; i++) o[(e * 4 + i) * 64] = t4[i];
        }
    }
    __syncthreads();
    // stores: piece pc = (tid >> 4) + 16 i of the item, 16 bytes per lane: a scalar base per i plus one per-thread offset
    const size_t cstride = (size_t)a.njt * a.nch * ND * 1024;            // bytes between the tiles of consecutive coefficients
    const int l16 = tid & 15, p0 = tid >> 4;
    const unsigned so = (unsigned)p0 * (unsigned)cstride + (unsigned)l16 * 16u;
    int8_t *dst = t.dst + (size_t)d * 1024;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const v4u o = *reinterpret_cast<const v4u *>(img + (p0 + 16 * i) * 64 + ((l16 ^ i) << 2));      // ((pc >> 4) & 7 = i: p0 < 16)
        v4u *p = reinterpret_cast<v4u *>(dst + (size_t)(16 * i) * cstride + so);
#ifdef SFG_AB
        if (a.fake & 2) p = reinterpret_cast<v4u *>(a.B + ((size_t)item * ND + d) * 32768 + (size_t)(i * 256 + tid) * 16);
#endif
        if (NT) __builtin_nontemporal_store(o, p); else *p = o;
    }
    __syncthreads();
}
// units u = 0 .. of the items item0, item0 + stride, ... < item_end (ND units per item), the loads of the next DEPTH - 1 units always in flight
template <int ND, int DEPTH, bool NT>
__device__ __forceinline__ void i8_move_run(const I8Args &a, unsigned item0, unsigned item_end, unsigned stride, unsigned *img, int tid) {
    if (item0 >= item_end) return;
    const unsigned nu = ((item_end - item0 + stride - 1) / stride) * ND;
#ifdef SFG_AB
    if (a.fake & 4) { for (unsigned u = 0; u < nu; u++) for (int i = 0; i < (a.fake >> 8); i++) __builtin_amdgcn_s_sleep(127); return; }       // hold the slot, move nothing
#endif
    v4u wa[8], wb[8], wc[8];
    const I8MoveLane l = i8_move_lane(a, tid);
    auto issue = [&](unsigned u, v4u (&w)[8]) { i8_move_issue<ND, NT>(a, l, item0 + (u / ND) * stride, (int)(u % ND), w); };
    auto finish = [&](unsigned u, v4u (&w)[8]) { i8_move_finish<ND, NT>(a, l, item0 + (u / ND) * stride, (int)(u % ND), w, img, tid); };
    // (past the last unit the prefetch re-reads it: a load behind a branch would make the counted waits inexact)
    auto upto = [&](unsigned u) { return u < nu ? u : nu - 1; };
    if constexpr (DEPTH == 3) {
        issue(0, wa); issue(upto(1), wb);
        for (unsigned u = 0;; u += 3) {
            issue(upto(u + 2), wc);
            finish(u, wa);
            if (u + 1 >= nu) break;
            issue(upto(u + 3), wa);
            finish(u + 1, wb);
            if (u + 2 >= nu) break;
            issue(upto(u + 4), wb);
            finish(u + 2, wc);
            if (u + 3 >= nu) break;
        }
    } else if constexpr (DEPTH == 2) {
        issue(0, wa);
        for (unsigned u = 0;; u += 2) {
            issue(upto(u + 1), wb);
            finish(u, wa);
            if (u + 1 >= nu) break;
            issue(upto(u + 2), wa);
            finish(u + 1, wb);
            if (u + 2 >= nu) break;
        }
    } else {
        for (unsigned u = 0; u < nu; u++) { issue(u, wa); finish(u, wa); }
    }
}
// mover workgroup mb of job.nblocks: its share of the job's items, the five-digit ones first
template <int DEPTH, bool NT>
__device__ __forceinline__ void i8_move_block(const MoveJob &job, unsigned mb, unsigned *img, int tid) {
    const unsigned lo = job.first, hi = job.first + job.count, nb = job.nblocks;
    const unsigned hi5 = hi < job.n5 ? hi : job.n5;
    if (lo < hi5) i8_move_run<5, DEPTH, NT>(job.a5, lo + mb, hi5, nb, img, tid);
    if (hi > job.n5) {
        const unsigned lo6 = lo > job.n5 ? lo : job.n5;
        // keep the round-robin phase across the boundary: block mb takes the items congruent to lo + mb modulo nb
        const unsigned ph = (lo + mb) % nb, r = lo6 % nb, start = lo6 + (ph + nb - r) % nb;
        i8_move_run<6, DEPTH, NT>(job.a6, start - job.n5, hi - job.n5, nb, img, tid);
    }
}
