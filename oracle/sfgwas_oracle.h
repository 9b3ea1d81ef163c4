/*
 * sfgwas_oracle.h — CPU restatement ("oracle") of the SF-GWAS local hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / the timed CPU baseline.  The product path (sfgwas_amd/, libsfgwas_hip.so)
 * never links or calls it.
 *
 * PARITY STATUS: **parity unpinned** against the Go binary.  The reference ships no tests,
 * golden vectors or fixtures and cannot be built here (no Go toolchain,
 * lattigo fork + mpc-core not vendored).  What pins this oracle instead:
 *   (i)   the integer kernels that are fully specified inside the reference
 *         (gwas/matmult.go:196-440 u128 MAC / REDC / MForm, :627-672 diagonals,
 *         :914-1505 BSGS schedule, gwas/filestream.go formats, mpc/beavermult.go:94-147)
 *         are restated 1:1 and checked against Python big-integer known answers
 *         (tests/test_oracle_*.py, tests/golden/);
 *   (ii)  everything that lives in the absent third-party modules
 *         (github.com/hcholab/lattigo/v2 v2.1.2-0.20230123224332-e8d68c24b94a: ring.NTT,
 *         ckks.EncoderBig.EncodeNTT, ckks.Evaluator.RotateNew key-switch;
 *         github.com/hhcho/mpc-core v0.0.0-20220828210829-24cf7abd1073: RElem arithmetic)
 *         is restated from the published algorithm and pinned by mathematical identities
 *         (NTT = evaluation at psi^(2*brev(i)+1); encode = exactly-rounded scaled inverse
 *         canonical embedding, checked against mpmath; rotate decrypts to the rotated
 *         vector; matmul decrypts to A*X).
 */
#ifndef SFGWAS_ORACLE_H
#define SFGWAS_ORACLE_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define ORC_MAXMOD 16

typedef struct orc_ring orc_ring;

/* ---- ring / modular arithmetic (lattigo ring package semantics; matmult.go:328,345,403) ---- */
/* moduli = nq ciphertext primes followed by np special primes. psi may be NULL (derived as
 * g^((q-1)/2N) for the smallest primitive root g, which is how lattigo derives it). */
orc_ring *orc_ring_new(int logN, int nq, int np, const uint64_t *moduli, const uint64_t *psi);
void orc_ring_free(orc_ring *r);
int orc_ring_N(const orc_ring *r);
uint64_t orc_ring_psi(const orc_ring *r, int mod);
uint64_t orc_ring_modulus(const orc_ring *r, int mod);
uint64_t orc_mred_params(uint64_t q);               /* q^-1 mod 2^64  (ring.MRedParams) */
void orc_bred_params(uint64_t q, uint64_t u[2]);     /* floor(2^128/q) -> {hi, lo} (ring.BRedParams) */
uint64_t orc_mform(uint64_t a, uint64_t q, const uint64_t u[2]); /* matmult.go:433-440 */
uint64_t orc_mred(uint64_t x, uint64_t y, uint64_t q, uint64_t qinv);
uint64_t orc_mulmod(uint64_t a, uint64_t b, uint64_t q);
uint64_t orc_powmod(uint64_t a, uint64_t e, uint64_t q);
uint64_t orc_invmod(uint64_t a, uint64_t q);
void orc_ntt(const orc_ring *r, int mod, uint64_t *a);   /* in place: natural -> bit-reversed evaluation order */
void orc_intt(const orc_ring *r, int mod, uint64_t *a);  /* inverse of orc_ntt */

/* ---- gwas/matmult.go integer kernels ---- */
/* c is an array of n {hi,lo} pairs (matmult.go:196-199 field order) */
void orc_mul_coeffs_and_add128(const uint64_t *a, const uint64_t *b, uint64_t *c_hilo, int n);        /* :247-289 */
void orc_reduce_and_add_uint128(const uint64_t *in_hilo, uint64_t *out, uint64_t qinv, uint64_t q, int n); /* :291-324 */
void orc_mform_vec(uint64_t *a, int n, uint64_t q);                                                    /* :411-431 */
void orc_canonical_reduce(uint64_t *a, int n, uint64_t q);                                             /* eval.Reduce at :358 */

/* ---- diagonals (matmult.go:627-672) ---- */
int orc_get_diag_bool(int r, int c, int dim, int index);
/* X: r x c int8 block, row stride ld. dst has dim entries. returns 1 if the diagonal exists */
int orc_get_diag(double *dst, const int8_t *X, size_t ld, int r, int c, int dim, int index);
void orc_rot_right(const double *v, double *out, int n, int nrot); /* convertToComplex128WithRot, real part */

/* ---- CKKS encode (lattigo ckks.EncoderBig.EncodeNTT restated; matmult.go:723) ----
 * coefficient-domain integers: coeffs[c] = round(scale * sigma^-1(v))_c, c < N, as signed 64-bit.
 * prec: 0 = __float128 (113-bit) arithmetic, 1 = double-double arithmetic */
void orc_encode_coeffs(const orc_ring *r, const double *slots_real, double scale, int64_t *coeffs, int prec);
/* NTT-domain plaintext for moduli 0..nlev-1, canonical residues (NOT Montgomery form). out[nlev][N] */
void orc_encode_ntt(const orc_ring *r, const double *slots_real, double scale, int nlev, uint64_t *out, int prec);

/* ---- rotations (crypto/basics.go:201-224 -> lattigo RotateNew restated) ---- */
typedef struct orc_rotkeys orc_rotkeys;
/* key material for one galois element: [beta][2][nq+np][N], NTT domain, normal (non-Montgomery) form */
orc_rotkeys *orc_rotkeys_new(const orc_ring *r);
void orc_rotkeys_free(orc_rotkeys *k);
void orc_rotkeys_set(orc_rotkeys *k, uint64_t galois_el, const uint64_t *key);
const uint64_t *orc_rotkeys_get(const orc_rotkeys *k, uint64_t galois_el);
int orc_rotkeys_beta(const orc_ring *r);
uint64_t orc_galois_for_rotation(const orc_ring *r, int k_left);   /* 5^k mod 2N */
void orc_automorphism_index(const orc_ring *r, uint64_t galois_el, uint32_t *index);
/* key switch of one NTT-domain poly cx at `level`; d0,d1: [level+1][N] */
void orc_keyswitch(const orc_ring *r, int level, const uint64_t *cx, const uint64_t *key, uint64_t *d0, uint64_t *d1);
/* ct layout: [2][level+1][N]. rotate LEFT by k slots (lattigo RotateNew(ct,k)). returns 0 ok, -1 missing key */
int orc_rotate_left(const orc_ring *r, const orc_rotkeys *keys, int level, const uint64_t *ct_in, int k, uint64_t *ct_out);
/* crypto.RotateRightWithEvaluator semantics (basics.go:201-210) */
int orc_apply_galois(const orc_ring *r, const orc_rotkeys *keys, int level, const uint64_t *ct_in, uint64_t galois_el, uint64_t *ct_out);   /* ConjugateNew: 2N-1 */
int orc_rotate_right(const orc_ring *r, const orc_rotkeys *keys, int level, const uint64_t *ct_in, int nrot, uint64_t *ct_out);

/* ---- input formats (scripts/plinkBedToBinary.py, filterMatrix.py) ---- */
int orc_bed_decode(const uint8_t *bed, size_t bed_bytes, size_t num_sample, size_t num_snp, int8_t *out);
void orc_filter_matrix(const int8_t *in, size_t nrows, size_t ncols, const uint8_t *rf, const uint8_t *cf, int8_t *out);

/* PLINK 2 .pgen hard calls (published PGEN specification restated; pinned by the reference's all.gcount.transpose.bin for record types 0 / 1) */
int orc_pgen_index(const uint8_t *f, size_t bytes, uint32_t *nv_out, uint32_t *ns_out, uint64_t **off, uint32_t **len, uint8_t **vrt);
int orc_pgen_decode_codes(const uint8_t *f, size_t bytes, uint32_t v0, uint32_t v1, uint8_t *genovec);
int orc_pgen_to_int8(const uint8_t *f, size_t bytes, uint32_t v0, uint32_t v1, const uint8_t *row_filter, const uint8_t *col_filter, int8_t *out);
int orc_pgen_geno_counts(const uint8_t *f, size_t bytes, const uint8_t *row_filter, uint32_t *counts);

/* ---- remaining evaluator ops of crypto/basics.go used between the matmuls (C2-C4) ---- */
void orc_ct_addsub(const orc_ring *r, int level, const uint64_t *a, const uint64_t *b, int sub, uint64_t *out);
void orc_mulrelin(const orc_ring *r, int level, const uint64_t *a, const uint64_t *b, const uint64_t *rlk, uint64_t *out);
void orc_mul_plain(const orc_ring *r, int level, const uint64_t *ct, const uint64_t *pt, uint64_t *out);
void orc_rescale(const orc_ring *r, int level, const uint64_t *ct, uint64_t *out /*[2][level][N]*/);
int orc_innersum_all(const orc_ring *r, const orc_rotkeys *keys, int level, const uint64_t *cts, int nct, uint64_t *out);
uint64_t orc_scale_up_exact(double value, double n, uint64_t q);
void orc_mul_const(const orc_ring *r, int level, const uint64_t *ct, double constant, uint64_t *out, double *scale_mult);
void orc_mul_const_and_add(const orc_ring *r, int level, const uint64_t *ct0, double scale0, double constant, uint64_t *out, double *scale_out);   /* parity unpinned */
void orc_add_const(const orc_ring *r, int level, const uint64_t *ct, double constant, double ct_scale, uint64_t *out);
void orc_add_plain(const orc_ring *r, int level, const uint64_t *ct, const uint64_t *pt, uint64_t *out);
void orc_gen_rlk(const orc_ring *r, const int8_t *s_coeff, uint64_t seed, uint64_t *key_out);

/* ---- test-side CKKS helpers (not on the reference hot path; used to build inputs / check outputs) ---- */
void orc_gen_secret(const orc_ring *r, uint64_t seed, int8_t *s_coeff /*N, ternary*/);
void orc_gen_rotkey(const orc_ring *r, const int8_t *s_coeff, uint64_t galois_el, uint64_t seed, uint64_t *key_out);
void orc_encrypt_coeffs(const orc_ring *r, const int8_t *s_coeff, int level, const int64_t *m_coeffs, uint64_t seed, uint64_t *ct_out);
/* phase = c0 + c1*s in coefficient domain, residues per modulus: out[level+1][N] */
void orc_decrypt_residues(const orc_ring *r, const int8_t *s_coeff, int level, const uint64_t *ct, uint64_t *out);
void orc_fill_uniform(const orc_ring *r, int level, uint64_t seed, uint64_t *ct_out /*[2][level+1][N]*/);

/* ---- the hot path: MatMult4Stream (matmult.go:1238-1505) ---- */
/* A: s x nbr ciphertexts, each [2][in_level+1][N]; geno: nrow x ncol int8 row-major (ld = ncol);
 * out: s x m_ct ciphertexts each [2][max_level][N] (level max_level-1), deterministic part only
 * (the reference adds it onto a fresh encryption of zero, matmult.go:1443).
 * sum/sqsum: ncol doubles or NULL.  enc_prec as in orc_encode_coeffs.  returns 0 on success. */
int orc_matmult4stream(const orc_ring *r, const orc_rotkeys *keys, double scale,
                       const uint64_t *A, int s, int in_level, int max_level,
                       const int8_t *geno, size_t nrow, size_t ncol,
                       int compute_sqsum, int square, int enc_prec,
                       uint64_t *out, double *sum, double *sqsum);

/* the same product in two phases (what a contraction-sharded multi-GPU run combines between the phases) */
int orc_matmult_accumulate(const orc_ring *r, const orc_rotkeys *keys, double scale, const uint64_t *A, int s, int in_level, int max_level,
                           const int8_t *geno, size_t nrow, size_t ncol, int square, int enc_prec, int b0, int b1,
                           uint64_t *acc_out /*[m_ct][d][s][2][L][N]*/, uint8_t *giant_active /*[d] or NULL*/);
int orc_matmult_finalize(const orc_ring *r, const orc_rotkeys *keys, int max_level, int s, int m_ct, const uint64_t *acc,
                         const uint8_t *giant_active, int g0, int g1, int accumulate, uint64_t *out);

/* MAC-only inner loop for cpu_baseline timing: reference loop order, u128 accumulators.
 * rot: [s][2][L][N] rotated ct (one baby), pt: [L][N] Montgomery-form plaintext, acc: [s][2][L][N]{hi,lo} */
void orc_cpmult_acc_v2(const uint64_t *rot, const uint64_t *pt_mont, uint64_t *acc_hilo, int s, int L, int N);

/* cpu_baseline leg of bench.py: the reference's MAC loop on nthreads host threads for ~seconds; returns MAC/s */
double orc_bench_mac(int s, int L, int N, int nthreads, double seconds, long long *macs_done);

/* the same MAC loop on the reference's data layout: shared rotCache[i][baby], u128 accCache[i][giant] of one block column,
 * workers taking whole diagonals (matmult.go:1065-1068,1121-1168) */
double orc_bench_mac_ref_layout(int s, int L, int N, int d, int nthreads, int items, int passes, long long *macs_done, int *threads_active,
                                double *seconds_out);

/* ---- DiagCache file format (gwas/filestream.go:19-282) ---- */
/* payload byte order for coefficients: big-endian u64 (lattigo ring.WriteCoeffsTo, unverified — see header) */
typedef struct orc_diagcache orc_diagcache;
orc_diagcache *orc_diagcache_create(const char *path, int d);
void orc_diagcache_set_tables(orc_diagcache *dc, const uint8_t *baby, const uint8_t *giant);
/* pv: vector_len pointers (NULL = empty) to [num_moduli][n] u64 */
int orc_diagcache_write(orc_diagcache *dc, const uint64_t *const *pv, int vector_len, int level, double scale, int n, int num_moduli, uint32_t shift);
orc_diagcache *orc_diagcache_open(const char *path, int d);
int orc_diagcache_header(const orc_diagcache *dc, uint64_t hdr[6], uint8_t *baby, uint8_t *giant);
/* reads next record; bufs: vector_len pointers to [num_moduli][n] storage; empty[i] set. returns 1 ok, 0 EOF */
int orc_diagcache_read(orc_diagcache *dc, uint64_t *const *bufs, uint8_t *empty, uint32_t *shift);
void orc_diagcache_close(orc_diagcache *dc);

/* ---- Beaver local products (mpc/beavermult.go:94-147), prime field with `limbs` 64-bit LE limbs ---- */
void orc_beaver_elem(int pid, int limbs, const uint64_t *mod, const uint64_t *ar, const uint64_t *am,
                     const uint64_t *br, const uint64_t *bm, uint64_t *out, size_t n);
void orc_beaver_matmul(int pid, int limbs, const uint64_t *mod, const uint64_t *ar, const uint64_t *am,
                       const uint64_t *br, const uint64_t *bm, uint64_t *out, int m, int k, int n);

/* ---- plaintext sketch + moments (gwas/pca.go:152-162; matmult.go:1292-1300) ---- */
void orc_sketch(const int8_t *X, size_t nrow, size_t ncol, const int32_t *bucket, const int8_t *sgn, int kp,
                double *sketch /*kp x ncol*/, uint64_t *xsum, uint64_t *x2sum);

/* f-4: fork-independent share algebra of MPC.SSToCMat (mpc/ss.go:84-110) */
void orc_ss_mask(int limbs, const uint64_t *mod, const uint64_t *bound, const uint64_t *rm, const uint64_t *rand, uint64_t *rm_masked, uint64_t *mask, size_t n);
void orc_ss_hub_share(int limbs, const uint64_t *mod, const uint64_t *revealed, const uint64_t *mask, uint64_t *share, size_t n);

/* ---- collective bootstrap, local work (mpc/mhe.go:222-348 -> lattigo v2.1.0 dckks/refresh.go; PARITY UNPINNED, see the .c file) ---- */
void orc_bigint_to_rns(const orc_ring *r, int nmod, const uint64_t *limbs /*[N][W] two's complement*/, int W, uint64_t *out /*[nmod][N]*/);
void orc_refresh_gen_shares(const orc_ring *r, int level, const uint64_t *ct, const uint64_t *sk, const uint64_t *crs, const uint64_t *mask, int W,
                            const int32_t *e0, const int32_t *e1, uint64_t *h0 /*[level+1][N]*/, uint64_t *h1 /*[nq][N]*/);
void orc_refresh_finish(const orc_ring *r, int level, const uint64_t *ct, const uint64_t *h0agg, const uint64_t *h1agg, const uint64_t *crs,
                        uint64_t *out /*[2][nq][N]*/);

/* the target-scale form the reference calls (mhe.go:315,330): lattigo v2.2.0 dckks/refresh.go restated, PARITY UNPINNED */
void orc_refresh_gen_shares_scaled(const orc_ring *r, int level, const uint64_t *ct, double ct_scale, double target_scale, const uint64_t *sk, const uint64_t *crs,
                                   const uint64_t *mask, int W, const int32_t *e0, const int32_t *e1, uint64_t *h0, uint64_t *h1);
void orc_refresh_finish_scaled(const orc_ring *r, int level, const uint64_t *ct, double ct_scale, double target_scale, const uint64_t *h0agg, const uint64_t *h1agg,
                               const uint64_t *crs, uint64_t *out);

/* splitmix64 — the synthetic-data PRNG shared by oracle, tests, bench and device generators */
uint64_t orc_splitmix64(uint64_t *state);

#ifdef __cplusplus
}
#endif
#endif
