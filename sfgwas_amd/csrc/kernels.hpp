// kernels.hpp — internal launch interface between the translation units of libsfgwas_hip.
#pragma once
#include "common.hpp"

// row r of a batch uses modulus m[r % period] (ciphertext rows, plaintext rows, key-switch rows are all periodic)
// m < 0 marks a row the kernel must leave untouched
constexpr int SFG_MAXPATTERN = 128;   // >= KSW_MAXDIG * SFG_MAXMOD
struct ModPattern { int period; int8_t m[SFG_MAXPATTERN]; };
// row r of a launch lives at base + (r / rpg) * gstride + (r % rpg) * N  (strides in words)
struct RowMap { int rpg; size_t gstride_in, gstride_out; };
// where the plaintext of diagonal `shift` (= shift0 + index in the batch) of block row g lands inside a panel that holds G
// block rows: slot ((shift / 91) * G + g) * 91 + shift % 91, i.e. [giant][g][baby] so that k = g*91 + baby is contiguous.
// G == 0: dense (slot = index in the batch)
struct PanelMap { int G, g, shift0; unsigned packed_mask = 0; int K = 0; };   // packed_mask bit l: rows of modulus l are written as packed-limb words (mac_dma.hip); bit 31: as digit planes instead (mac_i8.hip); bit 30: the 46-bit modulus too (six planes); bit 29: compact rows (every modulus in planes, a plaintext's planes back to back)
constexpr unsigned PT_COMPACT = 1u << 29;
// bit 28 (with bit 29): K-MAJOR panel - [column][plane][128-byte coefficient block][k < K][128 B]: the k rows of a column's coefficient block are adjacent, so a transposition
// unit reads 16 runs of 2 KiB instead of 256 runs of 128 B (PanelMap::K = rows per column; G == 0: plaintext p of the launch is column p / K, row p % K)
constexpr unsigned PT_KMAJOR = 1u << 28;

// ntt.hip
int launch_ntt_fwd(sfg_ctx *ctx, const u64 *in, u64 *out, size_t nrows, const ModPattern &pat);
int launch_ntt_inv(sfg_ctx *ctx, const u64 *in, u64 *out, size_t nrows, const ModPattern &pat);
int launch_ntt_plain(sfg_ctx *ctx, const double *pc, u64 *out, size_t nplain, int L);
struct MoveJob;                                   // i8_move.hpp: mover workgroups of the previous MAC launch's plaintext transposition riding in front of the NTT's
// The riding transposition (round 6): the plaintext panel of MAC launch k - 1 goes into the int8 MAC's tiles INSIDE the plaintext-NTT launches of launch k's encode
// (k_ntt_half3_move: `nblocks` mover workgroups first in every grid, one to a CU beside three NTT workgroups).  matmul.hip fills one per delayed MAC launch;
// launch_encode_rows deals its items out evenly over the NTT launches it makes and i8_ride_finish moves whatever is left.
struct PtRide;
int launch_ntt_plain_half(sfg_ctx *ctx, const double *pc, u64 *out_half, size_t nplain, int L, PanelMap pm, const uint32_t *perm = nullptr, const MoveJob *mv = nullptr);
int launch_expand_half(sfg_ctx *ctx, const u64 *half, u64 *full, size_t nrows);
int launch_ntt_fwd_map(sfg_ctx *ctx, const u64 *in, u64 *out, size_t nrows, const ModPattern &pat, const RowMap &rm);
int launch_ntt_inv_map(sfg_ctx *ctx, const u64 *in, u64 *out, size_t nrows, const ModPattern &pat, const RowMap &rm);
// mac.hip
struct MacStrides { size_t rot_k, rot_r, pt_k, pt_n, out_n, out_r; bool pt_half = false; bool pt_packed = false; bool pt_digits = false; bool i8 = false; bool i8_big = false; bool pt_digits_big = false;
                    const int8_t *B_small = nullptr, *B_big = nullptr; int kb = 0;
                    int pt_layout = 0, pt_L = 0;  // 1: digit-plane panel rows packed - a plaintext is its pt_L moduli's 5 / 6 planes of N/2 bytes back to back (208 KiB at PN14QP438, L = 5) instead of L rows of N/2 words (320 KiB), PanelMap bit 29; 2: K-major, bits 29 + 28 (mac_i8.hip i8_panel_rows)
                    int B_mode = 0;       // B_* given and B_mode 0: streamed tiles (k' = g * kb + baby); 1: the pass's layout, already transposed (riding mover): no pass; 2: the pass runs into these buffers
                    const int8_t *A_small = nullptr, *A_big = nullptr; };   // A_*: the transposed rot tiles of this launch are given (I8RotPre): no copy to look up or make   // B_*: the int8 MAC's plaintext tiles are already in place (streamed transposition, StagePack): no panel, k' = g * kb + baby   // i8: small moduli on the int8 MAC (mac_i8.hip); pt_digits: their panel rows hold five digit planes   // in words; pt_half: pt rows hold N/2 words (mirror-symmetric plaintexts)
int launch_mac_i8_small(sfg_ctx *ctx, const double *rotf, size_t rotf_k_stride, size_t rotf_r_stride, int plane0, const u64 *pt, u64 *out, int K, int R, int r0, int Ncols,
                        int l0, int nl, int accumulate, const MacStrides &st);       // mac_i8.hip
int launch_mac_i8_big(sfg_ctx *ctx, const double *rotf, size_t rotf_k_stride, size_t rotf_r_stride, int plane0, const u64 *pt, u64 *out, int K, int R, int r0, int Ncols,
                      int l0, int accumulate, const MacStrides &st);
size_t mac_i8_stream_bytes(int K, int nl, int ND, int copies_of_rot);
size_t mac_i8_tile_bytes(int Kp, int nl, int ND);      // plaintext tile buffer of nl moduli with ND digits, K' contraction steps (mac_i8.hip)
int launch_mac(sfg_ctx *ctx, const u64 *rot, const u64 *pt, u64 *out, int K, int R, int Ncols, int L, int accumulate);
int launch_mac_strided(sfg_ctx *ctx, const u64 *rot, const u64 *pt, u64 *out, int K, int R, int Ncols, int L, int accumulate, const MacStrides &st);
// encode.hip
int launch_skew(sfg_ctx *ctx, const int8_t *blk, size_t ld, int r, int c, int transposed, int square, int8_t *D);
// pcache (nullable): the block's slot of the plaintext coefficient cache, [8192 shifts][N/2] doubles.  mode 1: the FFT writes its rows there (and the NTT reads them);
// 2: the rows are there already (no FFT; D unused); 3: they are there in the block's other orientation (no FFT; NTT through the permutation table)
struct PcCache { double *slot = nullptr; int mode = 0; const uint32_t *perm = nullptr; };
// Streamed transposition (round 4): the plaintext NTT of a batch writes its digit planes DENSE into one small staging buffer (reused by every batch, so it lives in
// the Infinity Cache and never reaches HBM) and k_i8_pack_stage moves the batch into the int8 MAC's k-contiguous tiles on a second queue, beside the (fp64-issue
// bound) FFT of the next batch.  The 21 - 43 GB plaintext panel and its read + write pass per MAC launch disappear.
struct StagePack {
    u64 *stage = nullptr;                  // [batch][L][N/2 words]
    int8_t *Bs = nullptr, *Bb = nullptr;   // tiles of the small moduli [m][c][jt][ch][5][1 KiB] / of the 46-bit modulus [c][jt][ch][6][1 KiB]
    int g = 0, kb = 92, njt = 6, nch = 0;  // block row inside the MAC group: k' = g * kb + baby
    int l_big = -1, l_small0 = 0, n_small = 0;
    hipStream_t q = nullptr; hipEvent_t ev_ntt = nullptr, ev_pack = nullptr; bool pending = false;
    unsigned seq = 0;
};
// plaintexts per batch of the streamed transposition: cfg.stage_giants whole giant steps (default 11: 1001 plaintexts, 5005 NTT workgroups; SFG_STAGE_GIANTS)
int launch_i8_pack_stage(sfg_ctx *ctx, StagePack &sp, int shift_lo, int nshift, int L);      // mac_i8.hip
int launch_encode_rows(sfg_ctx *ctx, const int8_t *D, int shift0, int nshift, int L, u64 *pt, bool half_rows = false, int G = 0, int g = 0, unsigned packed_mask = 0, const PcCache *pcache = nullptr,
                       StagePack *sp = nullptr, PtRide *ride = nullptr);
int encode_rows_launches(const sfg_ctx *ctx, int nshift);          // NTT launches launch_encode_rows makes for nshift diagonals (panel form)
// mac_i8.hip: the riding transposition of one delayed MAC launch (panel of K = ng * 91 k-rows x 91 columns -> tile buffers mi8.Bs / mi8.Bb); ride.on stays false where the
// moduli are not one run of 35-bit ones plus at most one 46-bit one
int i8_ride_prepare(sfg_ctx *ctx, const u64 *panel, int K, int Ncols, size_t pt_k, size_t pt_n, int layout, int L, int launches, PtRide &ride);
int i8_ride_finish(sfg_ctx *ctx, PtRide &ride);                     // items no NTT launch took: a launch of mover workgroups alone on the current stream
int i8_ride_tiles(sfg_ctx *ctx, int K, int L, int8_t **Bs, int8_t **Bb);          // the two tile buffers (a launch that transposes by the pass uses them as well: B_mode 2)
// genoio.hip: dense int8 copy [nr][ld_out] of the stored sub-block (r0.., c0..) of a 2-bit packed matrix (c0 a multiple of 4)
int launch_geno_unpack(sfg_ctx *ctx, const sfg_geno *g, size_t r0, size_t c0, size_t nr, size_t nc, int8_t *out, size_t ld_out);
// rotate.hip
int launch_rotate_right(sfg_ctx *ctx, const u64 *in, u64 *out, int nct, int level, const int *nrot_host);
int launch_rotate_right_indexed(sfg_ctx *ctx, const u64 *in, int nin, u64 *out, int nct, int level, const int *nrot_host, const int *in_index);
int launch_rotate_right_indexed_f64(sfg_ctx *ctx, const u64 *in, int nin, double *outf, int nct, int level, const int *nrot_host, const int *in_index, int L,
                                    const size_t *out_slot = nullptr);
int launch_relinearize(sfg_ctx *ctx, const u64 *tmp, int nct, int level, const u64 *mid, u64 *out);
int launch_ct_add(sfg_ctx *ctx, const u64 *a, const u64 *b, u64 *out, size_t nct, int level);
// mac_dma.hip
int mac_dma_planes(sfg_ctx *ctx, int L, std::vector<int> &plane_of, std::vector<int> &is_big);
double mac_big_maxterm(u64 q);
int launch_rot_to_f64(sfg_ctx *ctx, const u64 *rot, size_t nrows, int nl_rot, int L, double *rotf);
int launch_mac_dma(sfg_ctx *ctx, const double *rotf, size_t rows_per_k, const u64 *pt, u64 *out, int K, int R, int Ncols, int L, int accumulate, const MacStrides &st,
                   const double *rotsum = nullptr);
int launch_rot_sum(sfg_ctx *ctx, const double *rotf, size_t rows_per_k, int K, int L, double *rotsum);
int launch_pack_pt(sfg_ctx *ctx, const u64 *in, u64 *out, size_t nrows, size_t words_per_row, int L, unsigned packed_mask);
unsigned mac_dma_packed_mask(sfg_ctx *ctx, int L);
bool mac_use_dma(const sfg_ctx *ctx);
// mac_bc.hip (same contract as launch_mac_dma; small-modulus plaintext rows packed)
int launch_mac_bc(sfg_ctx *ctx, const double *rotf, size_t rows_per_k, const u64 *pt, u64 *out, int K, int R, int Ncols, int L, int accumulate, const MacStrides &st,
                  const double *rotsum);
// matmul.hip: the baby-step rotation cache of block rows [b0, b1) in the MAC layout; tabs = per-row active-baby flags (null: all 91)
int rotcache_build_rows_tab(sfg_ctx *ctx, const u64 *A, int s, int in_level, int max_level, int nbr, int b0, int b1, const std::vector<std::vector<uint8_t>> *tabs, double *cache);
int sfg_diag_bool(int r, int c, int dim, int index);
// stream.hip: the call-wide rotation cache of an association scan (nullptr in *out = not applicable; caller hipFree()s)
int assoc_build_rotcache(sfg_ctx *ctx, const u64 *A_dev, int s, int in_level, int max_level, size_t nr, const std::vector<size_t> &widths, double **out);
// A caller's baby-step rotation cache held ONLY as the int8 MAC's transposed rot tiles (round 4): one pair of buffers (35-bit moduli / 46-bit modulus) per MAC group of
// G block rows.  The association scan multiplies one block column per batch against the cache of all its block rows: the fp64 operand form (1.86 GB per block row at
// s = 13, 115 GB at 500 000 samples) is then only the source of the tiles - built group by group in a scratch buffer and dropped - and the scan's MAC runs on the matrix core.
struct I8RotPre { int G = 0, nbr = 0, s = 0, L = 0; std::vector<int8_t *> As, Ab; };
int i8_rotpre_build(sfg_ctx *ctx, const u64 *A_dev, int s, int in_level, int max_level, int nbr, const std::vector<std::vector<uint8_t>> *tabs, size_t budget_bytes, const char *prefix, I8RotPre &pre);   // pre.G == 0 afterwards: not taken (fp64 path)
void i8_rotpre_free(I8RotPre &pre);
int matmul_resident_range_i8pre(sfg_ctx *ctx, const I8RotPre &pre, int s, int max_level, const sfg_geno *g, unsigned flags, int blk0, int blk1, uint64_t *out);
int matmul_accumulate_i8pre(sfg_ctx *ctx, const I8RotPre &pre, int s, int max_level, const sfg_geno *g, unsigned flags, int j0, int j1, int accumulate, uint64_t *acc,
                            size_t acc_col_words = 0);        // acc_col_words: words between consecutive block columns' accumulators (0: dense [column][91 giants]...)
int launch_i8_pack_rot_to(sfg_ctx *ctx, const double *rotf, size_t rotf_k_stride, size_t rotf_r_stride, int plane0, int K, int R, int l0, int nl, bool big, int8_t *A_out);      // mac_i8.hip
size_t mac_i8_rot_tile_bytes(int K, int nl, int ND);
// the rotation cache of an association scan in whichever form the context multiplies with
struct AssocRot { double *f64 = nullptr; I8RotPre pre; };
int assoc_build_rot(sfg_ctx *ctx, const u64 *A_dev, int s, int in_level, int max_level, size_t nr, const std::vector<size_t> &widths, AssocRot &out);
void assoc_free_rot(AssocRot &r);
int assoc_product(sfg_ctx *ctx, const AssocRot &r, const uint64_t *A_dev, int s, int in_level, int max_level, const sfg_geno *g, unsigned flags, int nct, uint64_t *out);

// stream.hip: the streamed association scan restricted to the batches k % nparts == part (fmt 0 = .bed, 1 = .pgen); see its definition
int assoc_stream_part(sfg_ctx *ctx, int fmt, const char *path, size_t num_sample, size_t num_snp, const uint8_t *row_filter, const uint8_t *col_filter,
                      size_t batch_snps, const uint64_t *A_dev, int s, int in_level, int max_level, unsigned flags,
                      uint64_t *out_dev, size_t out_ct_capacity, size_t *out_ct, double *sum_host, double *sqsum_host,
                      int part, int nparts, std::vector<std::pair<size_t, size_t>> *ranges);
