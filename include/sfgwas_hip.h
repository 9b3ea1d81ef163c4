/*
 * sfgwas_hip.h — C-ABI of libsfgwas_hip.so: the MI355X (gfx950) implementation of SF-GWAS's
 * per-party local linear-algebra hot path.
 *
 * The reference (hhcho/sfgwas, 100 % Go) has no FFI/plugin seam; the seam is cut at Go function level
 * (SURVEY.md §8b).  Each entry point below names the reference interface it replaces (file:line under
 * the reference tree); INTEGRATION.md shows the cgo stub a maintainer would add on the Go side.
 *
 * Conventions
 *  - plain pointers and sizes only; every function returns 0 on success, non-zero on error, and
 *    sfg_last_error(ctx) describes the failure (the reference panics on this path — matmult.go:361,
 *    filestream.go:60 — so the Go shim turns non-zero into panic()).
 *  - Ring: N = 2^logN (logN = 14 for the PN14QP438 preset used by the reference, gwas.go:169),
 *    nq ciphertext primes q_0..q_{nq-1} followed by np special primes; all < 2^47 and == 1 mod 2N
 *    (PN12..PN14 presets of gwas.go:164-177 in prime size; the 55-bit primes of PN15 / PN16 and logN != 14 are not supported).
 *  - Polynomial rows are N uint64 canonical residues in lattigo's NTT order
 *    (row[i] = p(psi^(2*bitrev(i)+1))); a ciphertext at level l is [2][l+1][N]
 *    (= ct.Value()[k].Coeffs[m][:] flattened, crypto.go:32-60).
 *  - "_dev" pointers are device (HBM) addresses on the context's GPU; "_host" are host pointers.
 *  - Threading: one host thread drives a context at a time - the exclusivity rule of the Go evaluator
 *    pool (crypto.go:311-316).  Concurrent callers (assoc.go:360-408 runs assoc_num_blocks_parallel MatMult4Stream
 *    calls at once) each take a fork (sfg_ctx_fork): forks share the immutable ring tables and key material of their
 *    parent and own their HIP queues, scratch, staging ring, timers and error string.  The library keeps no
 *    process-global state and reads the environment only inside sfg_ctx_create.
 */
#ifndef SFGWAS_HIP_H
#define SFGWAS_HIP_H
#include <stdint.h>
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct sfg_ctx sfg_ctx;
typedef struct sfg_geno sfg_geno;

/* ---- context: created after CollectiveInit (gwas.go:212), from cryptoParams.Params + ring tables ----
 * replaces: ring.NewRing(N, Qi) at matmult.go:328,345,403 and the evaluator/encoder pools (crypto.go:89-135).
 * psi[m]: the primitive 2N-th root lattigo uses for modulus m (ring.Ring.PsiMont un-Montgomery'd); NULL =>
 * derived from the smallest primitive root exactly as lattigo derives it. */
int sfg_ctx_create(sfg_ctx **out, int device, int logN, int nq, int np,
                   const uint64_t *moduli, const uint64_t *psi, double scale);
/* The configuration a deployment legitimately tunes - the counterpart of the reference's TOML keys (gwas/gwas.go:40-117 `Config`; SURVEY section 5: "GPU knobs = new
 * optional TOML keys; defaults must reproduce reference behaviour").  Zero in a field = the library's default.  None of these changes a result word.  The same fields can
 * be given by an operator through the environment (read once, at creation; it overrides this struct): SFG_MM_GROUP, SFG_MM_ACC_BUDGET_MB, SFG_ASSOC_ROTCACHE_MB,
 * SFG_KSW_BUDGET_MB, SFG_ENC_BATCH, SFG_UPLOAD_BLOCKING, SFG_MGPU_TRANSPORT, SFG_MGPU_CACHE_GB, SFG_RCCL_LIB.  Kernel A/B and diagnostic switches are NOT part of this
 * library: they exist in the experimenters' build only (`make -C sfgwas_amd/csrc ab`). */
typedef struct sfg_config {
    uint32_t struct_size;            /* sizeof(sfg_config) of the caller's header (sfg_config_default sets it): lets the struct grow */
    int mm_group;                    /* block rows of the matrix multiplied per MAC launch; 0 = 8, or 10 .. 24 when their operands fit the free HBM */
    size_t acc_budget_bytes;         /* accumulators of the block columns of one pass; 0 = 24 GiB */
    size_t assoc_rotcache_bytes;     /* largest call-wide baby-step rotation cache of an association scan; 0 = 160 GiB; SIZE_MAX = none (rebuilt per batch, as the reference does) */
    size_t ksw_budget_bytes;         /* key-switch scratch per job chunk; 0 = 4 GiB */
    int enc_batch;                   /* diagonals per encode FFT / plaintext-NTT launch pair, 64 .. 8192; 0 = 2048 */
    int upload_blocking;             /* != 0: blocking pointer-table uploads (needed under rocprofv3 --pmc) */
    const char *mgpu_transport;      /* sfg_mgpu_create*_ex: NULL / "rccl" (default) or "direct" (one process only: ranks read their peers' buffers) */
    size_t mgpu_cache_bytes;         /* sfg_mgpu_create*_ex: a rank's own Q'X^T rotation cache up to this size -> per-column pipelined exchange; 0 = 72 GiB; SIZE_MAX = never */
    const char *rccl_lib;            /* sfg_mgpu_create*_ex: library name to dlopen instead of librccl.so.1 */
} sfg_config;
void sfg_config_default(sfg_config *c);
int sfg_ctx_create_ex(sfg_ctx **out, int device, int logN, int nq, int np,
                      const uint64_t *moduli, const uint64_t *psi, double scale, const sfg_config *config /* NULL = defaults */);
/* a second caller on the same keys: replaces checking a private evaluator out of the pool (crypto.go:287-316,
 * ckks.NewEvaluator per goroutine at matmult.go:1110,1200,1371).  Load keys before forks run concurrently. */
int sfg_ctx_fork(sfg_ctx *parent, sfg_ctx **out);
void sfg_ctx_destroy(sfg_ctx *ctx);
const char *sfg_last_error(const sfg_ctx *ctx);     /* ctx may be NULL: error of a failed sfg_ctx_create */
int sfg_ctx_synchronize(sfg_ctx *ctx);
/* returns every scratch buffer the context has grown (plaintext panels, rotation operands, digit streams, accumulators) to the device, after its queues
 * have drained; the next call re-grows what it needs.  For callers that change to a very different product shape beside a large resident matrix.
 * (No reference counterpart: Go's garbage collector plays this role for the accCache / rotCache slices of matmult.go:1065-1129.) */
int sfg_ctx_release_scratch(sfg_ctx *ctx);
/* bytes the context currently keeps in scratch buffers whose name starts with `prefix` ("" = all of them; "assoc.rot8" / "assoc.rotf" = an association scan's
 * rotation cache as int8 tiles / fp64 rows, "mm." = the product's panels, operands and accumulators, "mi8." = the int8 MAC's streams).  Introspection for tests
 * and memory planning; sfg_malloc returns these buffers to the device by itself when a caller's allocation would otherwise fail between library calls. */
int sfg_ctx_scratch_bytes(const sfg_ctx *ctx, const char *prefix, size_t *bytes);
/* sfg_ctx_synchronize waits for every queue of the context (main, own, auxiliary key-switch queue).  It - like sfg_memcpy_d2h and the host-pointer product
 * entry points - FAILS while an encoder coefficient within 2^-50 of a rounding tie is outstanding (see sfg_ctx_encoder_near_ties): results whose
 * bit-exactness with the reference's 256-bit encoder cannot be proven do not leave the device silently. */
/* run all subsequent work of this context on the given hipStream_t, so that a caller can order other work (RCCL collectives, torch ops) against the
 * library.  The handle must be an explicit stream: NULL is HIP's default stream and is refused.  sfg_ctx_use_own_stream returns to the context's own
 * (non-blocking) queue. */
int sfg_ctx_set_stream(sfg_ctx *ctx, void *hip_stream);
int sfg_ctx_use_own_stream(sfg_ctx *ctx);

/* rotation key of one Galois element (cryptoParams.RotKs, crypto.go:50; generated at mhe.go:73, crypto.go:232-275).
 * key_host: [beta][2][nq+np][N], beta = ceil(nq/np), NTT domain; montgomery_form != 0 if the words are in
 * lattigo's Montgomery representation (they are in lattigo's SwitchingKey) */
int sfg_ctx_load_rotkey(sfg_ctx *ctx, uint64_t galois_el, const uint64_t *key_host, int montgomery_form);
int sfg_ctx_has_rotkey(const sfg_ctx *ctx, uint64_t galois_el);
/* copy a loaded key back to the host in normal (non-Montgomery) form, [beta][2][nq+np][N]: lets a CPU checker (bench.py's parity
 * gate) or a CPU-only party work with exactly the key material resident on the device */
int sfg_ctx_export_rotkey(sfg_ctx *ctx, uint64_t galois_el, uint64_t *key_host);
uint64_t sfg_galois_for_rotation(const sfg_ctx *ctx, int k_left);

/* ---- device memory (so ciphertexts / genotypes stay resident across calls) ---- */
int sfg_malloc(sfg_ctx *ctx, void **dev_ptr, size_t bytes);
int sfg_free(sfg_ctx *ctx, void *dev_ptr);
int sfg_memcpy_h2d(sfg_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int sfg_memcpy_d2h(sfg_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int sfg_memcpy_d2d(sfg_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes);      /* stream-ordered */

/* ---- ring substrate (lattigo ring.NTT / ring.InvNTT behind crypto/basics.go) ----
 * rows: nrows device rows of N words; mod_idx[r] selects the modulus (0..nq+np-1) of row r (host array) */
int sfg_ntt_rows(sfg_ctx *ctx, uint64_t *rows_dev, int nrows, const int *mod_idx_host);
int sfg_intt_rows(sfg_ctx *ctx, uint64_t *rows_dev, int nrows, const int *mod_idx_host);

/* ---- A1/A2/A4: the lazy MAC (matmult.go:247-399) as a batched modular GEMM ----
 * For every coefficient c < N and modulus l < L:
 *    out[n][r][l][c] (+)= sum_{k<K} rot[k][r][l][c] * pt[k][n][l][c]   mod q_l
 * rot: [K][R][L][N] rotated-ciphertext rows (R = 2*s: (i, poly)), pt: [K][Ncols][L][N] NTT-domain plaintexts
 * (canonical, NOT Montgomery form), out: [Ncols][R][L][N].  accumulate != 0 adds onto out.
 * Net effect of MulCoeffsAndAdd128 + ReduceAndAddUint128 + MForm + eval.Reduce: the canonical sum. */
int sfg_mac_dev(sfg_ctx *ctx, const uint64_t *rot_dev, const uint64_t *pt_dev, uint64_t *out_dev,
                int K, int R, int Ncols, int L, int accumulate);

/* Test hook (refused unless SFG_ENABLE_TEST_HOOKS=1 was set before sfg_ctx_create): the SAME sum through the kernels a product multiplies with by default -
 * the int8 matrix-core MAC of mac_i8.hip (five signed base-256 digits for moduli < 2^36, six for moduli < 2^47; transposition kernels, ring / cache kernel by
 * column count, Horner recombination, untile), reached exactly as sfg_matmul_* reaches it.  Those kernels serve the mirror-symmetric plaintexts a real slot
 * vector encodes to (P[N-1-x] = P[x]) from half rows: pt_half_dev is [K][Ncols][L][N/2] canonical words, the result is
 *    out[n][r][l][x] (+)= sum_k rot[k][r][l][x] * pt_half[k][n][l][min(x, N-1-x)]   mod q_l.
 * pt_form 0: the plaintext rows are first written as the digit planes the product's plaintext NTT emits (k_i8_pack_pt_digits); 1: as panel words - packed
 * limbs / plain words - the DiagCache product's form (k_i8_pack_pt).  Ncols <= 96, K < 21846.  Replaces nothing in the reference: it exists so that
 * matmult.go:247-324's arithmetic can be checked against the default kernels directly (tests/test_gpu_mac.py). */
int sfg_mac_i8_dev(sfg_ctx *ctx, const uint64_t *rot_dev, const uint64_t *pt_half_dev, uint64_t *out_dev,
                   int K, int R, int Ncols, int L, int accumulate, int pt_form);

/* ---- A6/A7: diagonal extraction + CKKS encode (matmult.go:636-731, EncodeNTT) ----
 * Encodes the generalized diagonals `shift` in [shift0, shift0+nshift) of one <= slots x slots int8 block
 * (block rows r, cols c, row stride ld; transposed != 0 reads the block transposed), each right-rotated by
 * d*(shift/d), into NTT-domain plaintexts for moduli 0..L-1: pt_dev[nshift][L][N] canonical residues.
 * Genotype values must already be cleaned (missing -> 0; squared if requested). */
int sfg_encode_diags_dev(sfg_ctx *ctx, const int8_t *block_dev, size_t ld, int r, int c, int transposed,
                         int shift0, int nshift, int L, uint64_t *pt_dev);
/* coefficient-domain result of the encoder for arbitrary real slot vectors (host convenience used by
 * Mask/MaskTrunc-style callers, basics.go:110-172): values_host[nvec][slots] -> coeffs_host[nvec][N] int64 */
int sfg_encode_coeffs_host(sfg_ctx *ctx, const double *values_host, int nvec, int64_t *coeffs_host);
/* Rounding audit of the encoder.  The reference rounds Delta * sigma^-1(v) computed with 256-bit big floats (NewEncoderBig(params, 256),
 * matmult.go:1019,1421); the device computes the same reals in double-double (better than 2^-58 absolute here).  The two can only round a
 * coefficient differently when its exact value lies within that distance of a tie; every coefficient within 2^-40 of a tie is counted
 * per context.  count == 0 after a call proves that call's plaintexts are the reference's, bit for bit; a non-zero count (expected
 * about once per 2 * 10^11 coefficients) is informational - the pipeline's own error is ~2^-59.  A coefficient within 2^-50 of a tie (about once per
 * 10^15) is NOT tolerated: every synchronising entry point fails until the counters are reset, and the affected products must be re-derived with a
 * big-float encoder on the host. */
int sfg_ctx_encoder_near_ties(sfg_ctx *ctx, unsigned long long *count, int reset);
/* Round 4: a genotype-diagonal coefficient inside the 2^-50 band is RE-DERIVED on the device before it can fail anything: p_j = (Delta/n) sum_t v_t cos(2 pi 5^t j / 2N)
 * is summed exactly - small integers times 100-bit fixed-point cosines, integer accumulation - so only the cosine table's error (< 2^-68 of a unit) is left, and a sum
 * farther than 2^-62 from the tie proves its rounding and replaces the double-double value (sfg_ctx_encoder_resolved counts them).  What stays unproven - closer than
 * 2^-62 (about once per 10^18 coefficients), a second such coefficient among the 16 - 20 one lane rounds, real-valued slot rows (sfg_encode_vectors_dev) - fails as
 * described.  Cost: 1.4 % of the encode FFT (the re-derivation is a non-inlined serial loop at the kernel's end; same-box A/B in profiles/r04_fft_resolver_ab.txt). */
int sfg_ctx_encoder_resolved(sfg_ctx *ctx, unsigned long long *count);
/* the coefficients within 2^-50 of a tie seen since the last reset that could NOT be proven (the condition that makes the synchronising entry points fail) */
int sfg_ctx_encoder_unprovable(sfg_ctx *ctx, unsigned long long *count);
/* With a plaintext coefficient cache on (sfg_geno_set_plaintext_cache) a cached row was audited when it was FILLED: "count == 0" then proves the plaintexts
 * of the calls since the cache was enabled.  When the 2^-50 condition is reported (a failing synchronising call) or reset, every cache the context owns
 * forgets its rows, so the recovery never keeps serving a row that was filled while the condition was outstanding. */
/* test hook for the failure path above: marks n coefficients as too close to a tie.  Refused unless the process set SFG_ENABLE_TEST_HOOKS=1 before
 * sfg_ctx_create (a production caller cannot wedge a context with it by accident). */
int sfg_ctx_encoder_inject_unsafe_for_test(sfg_ctx *ctx, unsigned long long n);
/* crypto.EncodeFloatVector (crypto.go:398-420; behind Mask / MaskTrunc / MaskWithScaling, basics.go:110-172, and
 * CPMult operands): nvec real slot vectors [nvec][slots] (host) -> NTT-domain plaintexts pt_dev[nvec][level+1][N]
 * at the context's default scale. The reference encodes at MaxLevel; a product at a lower level reads the first rows. */
int sfg_encode_vectors_dev(sfg_ctx *ctx, const double *values_host, int nvec, int level, uint64_t *pt_dev);

/* ---- C1/A8: rotations (crypto/basics.go:201-224 -> ckks.Evaluator.RotateNew) ----
 * batch of nct ciphertexts at `level`, each [2][level+1][N]; ct j is rotated RIGHT by nrot_host[j]
 * (RotateRightWithEvaluator semantics: nrot mod slots, 0 = copy). in/out may not alias. */
int sfg_rotate_right_dev(sfg_ctx *ctx, const uint64_t *ct_in_dev, uint64_t *ct_out_dev, int nct, int level,
                         const int *nrot_host);
/* eval.ConjugateNew behind crypto.ComplexConjugate / CReal (basics.go:826-846): the automorphism X -> X^galois_el with the switching key loaded
 * under that element by sfg_ctx_load_rotkey (lattigo's conjugation key: galois element 2N-1).  Batch of nct ciphertexts; in/out may not alias. */
int sfg_ct_galois_dev(sfg_ctx *ctx, const uint64_t *ct_in_dev, uint64_t *ct_out_dev, int nct, int level, uint64_t galois_el);
/* C4: element-wise ciphertext add (eval.Add, basics.go:174, matmult.go:1225,1494): out = a + b */
int sfg_ct_add_dev(sfg_ctx *ctx, const uint64_t *a_dev, const uint64_t *b_dev, uint64_t *out_dev, int nct, int level);
/* C4: eval.Sub (crypto.CSub, basics.go:575-590): out = a - b */
int sfg_ct_sub_dev(sfg_ctx *ctx, const uint64_t *a_dev, const uint64_t *b_dev, uint64_t *out_dev, int nct, int level);

/* C6: crypto.DropLevel -> eval.DropLevelNew (basics.go:806-824; FlattenLevels :514-531 drops a matrix to its minimum level):
 * in [nct][2][level_in+1][N] -> out [nct][2][level_out+1][N].  level_out == level_in copies (CopyNew); level_out > level_in fails. */
int sfg_ct_drop_level_dev(sfg_ctx *ctx, const uint64_t *in_dev, uint64_t *out_dev, int nct, int level_in, int level_out);

/* C4: eval.MultByConst / AddConst / AddNew(ct, plaintext) behind crypto.CMultConst, CMultConstRescale, AddConst, CAddConst,
 * AddPlain, CPAdd (basics.go:183-199, 472-497, 533-551, 592-611). scalars_host[level+1]: one canonical residue per modulus
 * (lattigo scaleUpExact(constant, scale, q_m); the host side owns that rule and the scale bookkeeping). out may alias ct. */
int sfg_ct_mul_scalar_dev(sfg_ctx *ctx, const uint64_t *ct_dev, const uint64_t *scalars_host, uint64_t *out_dev, int nct, int level);
/* eval.MultByConstAndAdd (pca.go:264; qrfact.go:195,280): acc += ct * scalars[m], both polynomials, acc and ct at the same level.  The scale
 * matching lattigo performs first (MultByConst of the receiver by floor(scale ratio), SetScale) is host-side: crypto::MultByConstAndAddDev. */
int sfg_ct_mul_scalar_add_dev(sfg_ctx *ctx, const uint64_t *ct_dev, const uint64_t *scalars_host, uint64_t *acc_dev, int nct, int level);
int sfg_ct_add_scalar_dev(sfg_ctx *ctx, const uint64_t *ct_dev, const uint64_t *scalars_host, uint64_t *out_dev, int nct, int level);
int sfg_ct_add_plain_dev(sfg_ctx *ctx, const uint64_t *ct_dev, const uint64_t *pt_dev, size_t pt_stride, uint64_t *out_dev,
                         int nct, int level);

/* ---- C3: ciphertext products (crypto.CMult / CPMult / Mask, basics.go:110-172, 386-470) ----
 * The relinearisation key (cryptoParams.Rlk, crypto.go:47) has the same layout as a rotation key:
 * [beta][2][nq+np][N], montgomery != 0 if in lattigo's stored Montgomery form. */
int sfg_ctx_load_relinkey(sfg_ctx *ctx, const uint64_t *key_host, int montgomery);
/* eval.MulRelinNew(a, b): degree-2 tensor product + relinearisation, batch of nct pairs; level unchanged, scale = product.
 * No rescale (call sfg_ct_rescale_dev, as CMult does at basics.go:394). out may alias neither input. */
int sfg_ct_mulrelin_dev(sfg_ctx *ctx, const uint64_t *a_dev, const uint64_t *b_dev, uint64_t *out_dev, int nct, int level);
/* eval.MulRelinNew(plaintext, ct): both polynomials times an NTT-domain plaintext [level+1][N]; ct j uses
 * pt_dev + j*pt_stride words (pt_stride 0 = one plaintext for all, e.g. a Mask). out may alias ct. */
int sfg_ct_mul_plain_dev(sfg_ctx *ctx, const uint64_t *ct_dev, const uint64_t *pt_dev, size_t pt_stride, uint64_t *out_dev,
                         int nct, int level);
/* one step of eval.Rescale = ring.DivRoundByLastModulusNTT on both polynomials: in [nct][2][level+1][N] ->
 * out [nct][2][level][N] (level-1); the caller divides the scale by q_level. Fails at level 0 like lattigo. */
int sfg_ct_rescale_dev(sfg_ctx *ctx, const uint64_t *in_dev, uint64_t *out_dev, int nct, int level);
/* C2: crypto.InnerSumAll (basics.go:278-292): sum of the nct ciphertexts and of all their slots, in every slot of the one
 * output ciphertext. Needs the rotation keys for left rotations by 1,2,4,..,slots/2. */
int sfg_ct_innersum_dev(sfg_ctx *ctx, const uint64_t *in_dev, int nct, int level, uint64_t *out_dev);

/* ---- F1 + A11: genotype matrix residency (replaces GenoFileStream reads + the DiagCache of
 * MatMult4StreamPreprocess, matmult.go:914-1041; filestream.go:284-494).
 * geno: nrow x ncol int8 row-major (row stride ld), -1 = missing.  The handle keeps ONE int8 copy in HBM
 * that serves both X (transpose = 0) and X^T (transpose = 1) products. */
int sfg_geno_upload(sfg_ctx *ctx, const int8_t *geno_host, size_t nrow, size_t ncol, size_t ld, sfg_geno **out);
int sfg_geno_from_device(sfg_ctx *ctx, const int8_t *geno_dev, size_t nrow, size_t ncol, size_t ld, sfg_geno **out);
void sfg_geno_free(sfg_ctx *ctx, sfg_geno *g);
/* Row-streamed registration.  The reference never holds its matrix: MatMult4StreamPreprocess (gwas/matmult.go:914-1041) pulls ONE ROW at a time out of
 * GenoFileStream.NextRow (gwas/filestream.go:414-426).  sfg_geno_create makes the resident nrow x ncol matrix (contents undefined until written),
 * sfg_geno_write_rows copies rows [row0, row0 + nrows) from a host chunk (row stride ld; the call returns when the chunk may be refilled), so a caller needs a
 * staging buffer of a few thousand rows and nothing that scales with nrow * ncol.  sfg_geno_compare_rows compares a chunk of the rows of the matrix arriving now
 * with the RESIDENT matrix `g` viewed with `flags` (0, or SFG_TRANSPOSE: the rows of its transpose) on the device, entry by entry, and ADDS the number of
 * differing entries to *ndiff: that is how the second registration of pca.go:112-113 (X, then X^T from its own file) is recognised as a view of the first
 * without X^T being held anywhere - exactly, not by a hash.  sfg_pinned_alloc / _free: page-locked host memory for the staging buffer (optional). */
int sfg_geno_create(sfg_ctx *ctx, size_t nrow, size_t ncol, sfg_geno **out);
int sfg_geno_write_rows(sfg_ctx *ctx, sfg_geno *g, size_t row0, size_t nrows, const int8_t *rows_host, size_t ld);
int sfg_geno_compare_rows(sfg_ctx *ctx, const sfg_geno *g, unsigned flags, size_t row0, size_t nrows, const int8_t *rows_host, size_t ld, uint64_t *ndiff);
int sfg_pinned_alloc(sfg_ctx *ctx, void **host_ptr, size_t bytes);
int sfg_pinned_free(sfg_ctx *ctx, void *host_ptr);
/* Plaintext coefficient cache of a resident matrix - the device-memory counterpart of the reference's on-disk DiagCache (MatMult4StreamPreprocess,
 * gwas/matmult.go:1228-1334, read back per iteration at :1386-1400).  After this call the products over `g` keep, per 8192 x 8192 block (and per SFG_SQUARE
 * flavour), the encoder's rounded coefficient rows (512 MB) until max_bytes are held; a later product over a cached block - in EITHER orientation: the
 * diagonals of the transposed block are slot rotations of the cached ones, i.e. automorphism images of the cached plaintexts - skips the skew and the FFT and
 * produces the same words.  max_bytes = 0 drops the cache.  The matrix must not change while cached; only `ctx` may multiply with `g` meanwhile.
 * sfg_geno_plaintext_cache_stats: blocks held, bytes held, block encodes served from the cache, blocks filled (any pointer may be NULL). */
int sfg_geno_set_plaintext_cache(sfg_ctx *ctx, const sfg_geno *g, size_t max_bytes);
int sfg_geno_plaintext_cache_stats(sfg_ctx *ctx, const sfg_geno *g, size_t *blocks, size_t *bytes, size_t *hits, size_t *fills);
/* Input pipeline on the device (replaces the per-batch shell-outs of assoc.go:389 to the Python converters under scripts/):
 * scripts/plinkBedToBinary.py + filterMatrix.py: `bed_host` is a whole SNP-major PLINK .bed image (3 magic bytes +
 * ceil(num_sample/4) bytes per SNP; codes 00->2, 01->-1, 10->1, 11->0); rows (samples) / columns (SNPs) whose filter
 * byte is zero are dropped (nullptr = keep all). Result: resident sample-major int8 matrix. Fails on a size mismatch
 * (the script's assert) or wrong magic. */
int sfg_geno_from_bed(sfg_ctx *ctx, const uint8_t *bed_host, size_t bed_bytes, size_t num_sample, size_t num_snp,
                      const uint8_t *row_filter, const uint8_t *col_filter, sfg_geno **out);
/* PLINK 2 .pgen hard calls (the reference's shipped example data, config 1): replaces gwas/utilities.go:141 FilterMatrixFilePgen ->
 * scripts/filterMatrixPgen.sh:12-18 (plink2 --pfile --keep --extract --indiv-sort none --make-bed, then plinkBedToBinary.py) for variants [v0, v1)
 * of one file (v1 = 0: all).  pgen_host: the whole .pgen image (storage mode 0x10 or 0x02; hard calls only).  row_filter: one byte per sample of the
 * .psam (the --keep list as a mask), col_filter: one byte per variant of [v0, v1) (the --extract list), zero = drop, NULL = keep all.
 * Result: resident sample-major int8 (ALT allele count, missing = -1), i.e. the bytes of the temporary file GenoFileStream would read. */
int sfg_pgen_dims(sfg_ctx *ctx, const uint8_t *pgen_host, size_t pgen_bytes, size_t *num_sample, size_t *num_variant);
int sfg_geno_from_pgen(sfg_ctx *ctx, const uint8_t *pgen_host, size_t pgen_bytes, size_t v0, size_t v1,
                       const uint8_t *row_filter, const uint8_t *col_filter, sfg_geno **out);
/* scripts/preprocessing/computeGenoCounts.py (plink2 --keep --geno-counts, columns 5-10; the file behind geno_count_file, read at
 * gwas/qualcontrol.go:595): counts_host[6][num_variant] uint32 = HOM_REF_CT, HET_REF_ALT_CTS, TWO_ALT_GENO_CTS, HAP_REF_CT, HAP_ALT_CTS, MISSING_CT */
int sfg_pgen_geno_counts(sfg_ctx *ctx, const uint8_t *pgen_host, size_t pgen_bytes, const uint8_t *row_filter, uint32_t *counts_host);
/* 2-bit packed residency (c4 is 25 GB instead of 100 GB; c5 fits 8 GPUs): codes 0, 1, 2 = the genotype, 3 = missing, 4 columns per byte.
 * Every product entry point takes a packed handle (blocks are expanded on the fly, ~1 % of a block's time); results are bit-identical.  Fails if a
 * value above 2 is present (the int8 layout stays available for such matrices).  Free the int8 handle afterwards to release its memory. */
int sfg_geno_pack(sfg_ctx *ctx, const sfg_geno *g, sfg_geno **out);
int sfg_geno_unpack(sfg_ctx *ctx, const sfg_geno *g, sfg_geno **out);
int sfg_geno_dims(const sfg_geno *g, size_t *nrow, size_t *ncol);
/* host [nrow][ncol] copy of a resident matrix (interoperability with CPU-only parties, tests) */
int sfg_geno_download(sfg_ctx *ctx, const sfg_geno *g, int8_t *host);
/* scripts/transposeMatrix.py as a materialised copy (the products themselves use SFG_TRANSPOSE on the one copy) */
int sfg_geno_transpose(sfg_ctx *ctx, const sfg_geno *g, sfg_geno **out);
/* scripts/mergeMatrices.py: column-wise concatenation of k resident matrices with equal row counts */
int sfg_geno_concat_cols(sfg_ctx *ctx, const sfg_geno *const *parts, int k, sfg_geno **out);
/* P2: per-column sum / sum of squares after missing->0 (matmult.go:1292-1300); either may be NULL */
int sfg_geno_colsums(sfg_ctx *ctx, const sfg_geno *g, double *sum_host, double *sqsum_host);

/* ---- A9/A10: the full product (MatMult4Stream matmult.go:1238-1505, MatMult4StreamCompute :1043-1236) ----
 * A: s x nbr ciphertexts [s][nbr][2][in_level+1][N] (nbr = ceil(rows/slots) of the operand orientation);
 * out: s x m_ct ciphertexts [s][m_ct][2][max_level][N] at level max_level-1, scale = A.scale * Params.Scale().
 * The result is the deterministic sum over giant steps; the reference adds it onto a fresh encryption of zero
 * (CZeroMat, matmult.go:1174,1443) which the Go shim keeps doing.
 * flags: SFG_SQUARE squares genotypes after missing->0 (matmult.go:1301-1303); SFG_TRANSPOSE multiplies by X^T. */
#define SFG_SQUARE 1u
#define SFG_TRANSPOSE 2u
#define SFG_STREAM_DIRECT 4u      /* sfg_assoc_stream_bed only: read the file with O_DIRECT (4096-byte aligned ranges, no page cache) */
int sfg_matmul_resident_dev(sfg_ctx *ctx, const uint64_t *A_dev, int s, int in_level, int max_level,
                            const sfg_geno *g, unsigned flags, uint64_t *out_dev);
/* MatMult4StreamCompute on the reference's OWN on-disk cache (matmult.go:1043-1236 reading the DiagCache files that
 * MatMult4StreamPreprocess of a CPU party wrote, filestream.go:19-282): files <prefix>_<bi>.bin for bi < nbr hold NTT + Montgomery-form
 * plaintexts as big-endian words; they are streamed, converted on the device and multiplied without re-encoding.  out: s x vectorLen
 * ciphertexts, as sfg_matmul_resident_dev.  Fails like the reference when a file is missing (os.Open panics, filestream.go:59-61).
 * Header fields are validated against the ring and the file size before anything is allocated.  Records are expected in the order
 * MatMult4StreamPreprocess writes them (increasing shift, matmult.go:1001-1035): the records of one giant step then form ONE MAC launch; any
 * other order still gives the right sums, at one upload + launch per change of giant step. */
int sfg_matmul_from_cache(sfg_ctx *ctx, const uint64_t *A_dev, int s, int in_level, int max_level, const char *cache_prefix, int nbr, uint64_t *out_dev);
/* MatMult4StreamPreprocess with its on-disk result (gwas/matmult.go:914-1041): writes <prefix>_<bi>.bin for every block row of the (optionally transposed)
 * resident matrix in the reference's DiagCacheStream format (filestream.go:144-231: 6 x u64 LE header, 2 d table bytes, per active shift u64 LE length,
 * u32 LE shift, per block column u8 isEmpty + (max_level + 1) x N words, NTT + Montgomery form, big-endian) from diagonals encoded ON THE DEVICE - for
 * interoperability with CPU-only parties and for sfg_matmul_from_cache.  Existing files are kept, as the reference keeps them (:928-931); records are in
 * increasing shift (the reference's worker pool delivers them in nearly that order; every reader takes them in any order).  96 bytes per genotype:
 * a 10 000 x 100 000 matrix makes 96 GB - the resident int8 matrix is the default for a reason.  *files_written (optional): files created by this call. */
int sfg_diagcache_write(sfg_ctx *ctx, const sfg_geno *g, unsigned flags, int max_level, const char *cache_prefix, int *files_written);
/* header of <prefix>_<block_row>.bin: {vectorLen (= block columns), level, scale (f64 bits), n, numModuli, rowSize} */
int sfg_diagcache_header(sfg_ctx *ctx, const char *cache_prefix, int block_row, uint64_t hdr[6]);
/* host-pointer form, one call = MatMult4Stream(cps, A, gfs, maxLevel, computeSquaredSum, square, nproc) */
int sfg_matmul_stream(sfg_ctx *ctx, const uint64_t *A_host, int s, int in_level, int max_level,
                      const int8_t *geno_host, size_t nrow, size_t ncol, size_t ld, unsigned flags,
                      uint64_t *out_host, double *sum_host, double *sqsum_host);
/* ---- f-3: the association scan's batches streamed from storage (gwas/assoc.go:340-420, GenoBlockMult; BASELINE config 5) ----
 * Replaces, per batch of `batch_snps` KEPT SNPs: FilterMatrixFilePgen (plink2 and the Python converters under scripts/ writing an int8 temp file), NewGenoFileStream,
 * MatMult4Stream(cps, mat, X, 5, false, square, nproc) and the final crypto.ConcatCipherMatrix.  `bed_path` is a SNP-major PLINK .bed of
 * num_sample x num_snp (a .pgen is converted once with plink2 --make-bed); row_filter / col_filter: one byte per sample / SNP, zero = drop, NULL = keep
 * all.  A batch is one contiguous byte range of the file: read by a reader thread into pinned memory while the GPU multiplies the previous batch,
 * moved to HBM as packed 2-bit codes, decoded / filtered / transposed on the device.  A_dev: [s][ceil(kept_samples / slots)] ciphertexts;
 * out_dev: [s][out_ct_capacity][2][max_level][N]; *out_ct = sum over batches of ceil(kept / slots).  sum_host / sqsum_host (optional):
 * [*out_ct * slots] column sums in the reference's padded layout (dosageSum[outShift + c], assoc.go:404-405).  flags: SFG_SQUARE, SFG_STREAM_DIRECT.
 * The baby-step rotation cache of `A` (rotCache[i][baby], matmult.go:1373-1377 - a function of A alone, rebuilt by the reference in every MatMult4Stream
 * call) is built ONCE per call and shared by all batches when it fits (SFG_ASSOC_ROTCACHE_MB, 0 = per batch): as the int8 MAC's rot tiles where every
 * modulus multiplies on the matrix core (1.3 GB per block row for s <= 15), else as fp64 operand rows (1.86 GB per block row at s = 13; SFG_ASSOC_I8=0).
 * The cache and the call's staging buffers (two pinned file slots of one batch each, the decoded batches) stay in the context's scratch pools for the next
 * call of the scan (sfg_ctx_scratch_bytes(ctx, "assoc.", ..); sfg_ctx_release_scratch returns them).
 * A Go shim that keeps calling per batch gets the same saving from sfg_rotcache_build_rows_dev + sfg_matmul_resident_range_rc_dev. */
int sfg_assoc_stream_bed(sfg_ctx *ctx, const char *bed_path, size_t num_sample, size_t num_snp, const uint8_t *row_filter, const uint8_t *col_filter,
                         size_t batch_snps, const uint64_t *A_dev, int s, int in_level, int max_level, unsigned flags,
                         uint64_t *out_dev, size_t out_ct_capacity, size_t *out_ct, double *sum_host, double *sqsum_host);
/* the same scan over one chromosome's .pgen image - BASELINE config 1's path (assoc.go:371-416 with isPgen): per batch of `batch_snps` kept variants
 * FilterMatrixFilePgen (plink2 + plinkBedToBinary.py) + MatMult4Stream, decoded natively on the device (sfg_geno_from_pgen).  row_filter: one byte per
 * sample (the --keep list), col_filter: one byte per variant of the file (snpFilt).  Layout of out_dev / sum_host / sqsum_host as above. */
int sfg_assoc_pgen(sfg_ctx *ctx, const uint8_t *pgen_host, size_t pgen_bytes, const uint8_t *row_filter, const uint8_t *col_filter, size_t batch_snps,
                   const uint64_t *A_dev, int s, int in_level, int max_level, unsigned flags,
                   uint64_t *out_dev, size_t out_ct_capacity, size_t *out_ct, double *sum_host, double *sqsum_host);
/* ... and straight from the .pgen on disk (config 5: 10 M SNPs per party, the file does not fit host memory as one image): the header tables are read once,
 * every batch is the contiguous byte range of its variant records (plus the LD base its first records may need), read ahead by a reader thread into pinned
 * memory (SFG_STREAM_DIRECT: O_DIRECT) while the GPU works on the previous batch.  Sample and variant counts come from the file's header. */
int sfg_assoc_stream_pgen(sfg_ctx *ctx, const char *pgen_path, const uint8_t *row_filter, const uint8_t *col_filter, size_t batch_snps,
                          const uint64_t *A_dev, int s, int in_level, int max_level, unsigned flags,
                          uint64_t *out_dev, size_t out_ct_capacity, size_t *out_ct, double *sum_host, double *sqsum_host);
/* sharding hooks for one-process-per-GPU runs: restrict a resident product to block columns
 * [j0, j1) of the output (X: SNP-column blocks) or block rows [b0, b1) of the contraction (X^T) */
int sfg_matmul_resident_range_dev(sfg_ctx *ctx, const uint64_t *A_dev, int s, int in_level, int max_level,
                                  const sfg_geno *g, unsigned flags, int blk0, int blk1, uint64_t *out_dev);
/* two-phase form of the same product, for contraction-sharded (X^T) multi-GPU runs.  Key switching is not
 * bit-linear, so partial sums must be combined BEFORE the giant-step rotations to stay bit-exact with the reference:
 *   accumulate: acc[(j-j0)][giant < 91][i < s][2][max_level][N] (+)= sum over operand block rows [b0,b1) and baby steps
 *   (ranks then reduce-scatter / all-reduce acc as uint64 and call sfg_reduce_rows_dev)
 *   finalize:   out[i][j][2][max_level][N] (+)= sum_{giant in [g0,g1)} RotateRight(acc[j][giant][i], -giant*91)
 *               (matmult.go:1443-1502); out has ncolb block columns */
int sfg_matmul_accumulate_dev(sfg_ctx *ctx, const uint64_t *A_dev, int s, int in_level, int max_level,
                              const sfg_geno *g, unsigned flags, int b0, int b1, int j0, int j1,
                              int accumulate, uint64_t *acc_dev);
int sfg_matmul_finalize_dev(sfg_ctx *ctx, const uint64_t *acc_dev, int s, int max_level, int ncolb,
                            int g0, int g1, int accumulate, uint64_t *out_dev);
/* finalize for a rank of a giant-sharded run: acc_dev is [ncolb][acc_giants][s][2][max_level][N] and slot g of a block column holds
 * giant step giant_base + g (what a reduce-scatter over the giant axis leaves on each rank); slots [g0, g1) are aligned, giant steps
 * >= 91 ignored */
int sfg_matmul_finalize_slots_dev(sfg_ctx *ctx, const uint64_t *acc_dev, int s, int max_level, int ncolb, int acc_giants, int giant_base,
                                  int g0, int g1, int accumulate, uint64_t *out_dev);
/* ---- the baby-step rotation cache as an object (SURVEY.md §8e "builds the full rotation cache (or the rotation cache is built once and
 * broadcast)"): rotCache[i][baby] = RotateRightWithEvaluator(A[i][bi], -baby) of matmult.go:1083-1119,1373-1377 in the MAC's fp64 operand layout
 *     cache[bi - row0][baby < 91][i < s][poly < 2][rowf doubles]   followed by tail_doubles zeros,   job_doubles = 91 * 2 * rowf.
 * Every rank of the output-sharded product Q*X multiplies by the cache of ALL operand block rows; rank r key-switches only the inputs
 * (bi, i) of its job range (job = bi*s + i: the decomposition of an input is shared by its 91 rotations) into job-major staging
 *     staged[job - job0][baby][poly][rowf],
 * the ranks all-gather the staging buffers and scatter them into the cache layout.  All 91 baby steps are rotated (rotations the reference's
 * active-baby table would skip meet zero plaintexts only), so every baby rotation key must be loaded (crypto.go:252-263 generates them all).
 * The MAC keeps a transposed copy of the operand it last multiplied with, keyed by the cache pointer, its shape and strides, and a PER-CONTEXT generation
 * count that sfg_rotcache_scatter_dev / sfg_rotcache_build_rows_dev advance on the context they run on.  A cache buffer written by any other route - a
 * collective straight into the layout, a device copy of a saved cache, a build on ANOTHER (forked) context - must be announced to the multiplying context
 * with sfg_rotcache_invalidate before its next *_rc_dev product, or that product may read the stale copy. */
int sfg_rotcache_invalidate(sfg_ctx *ctx);
int sfg_rotcache_layout(sfg_ctx *ctx, int s, int max_level, size_t *job_doubles, size_t *tail_doubles);
int sfg_rotcache_build_jobs_dev(sfg_ctx *ctx, const uint64_t *A_dev, int s, int in_level, int max_level, int nbr, int job0, int job1, double *staged_dev);
/* staged jobs [job0, job1) -> cache rows [row0, row0 + nrows); also zeroes the tail */
int sfg_rotcache_scatter_dev(sfg_ctx *ctx, const double *staged_dev, int s, int max_level, int job0, int job1, int row0, int nrows, double *cache_dev);
/* the cache of block rows [b0, b1) written in place (a rank's own rows of the contraction-sharded product Q'*X^T) */
int sfg_rotcache_build_rows_dev(sfg_ctx *ctx, const uint64_t *A_dev, int s, int in_level, int max_level, int nbr, int b0, int b1, double *cache_dev);
/* sfg_matmul_resident_range_dev / sfg_matmul_accumulate_dev on a prebuilt cache that covers exactly the operand block rows the call contracts
 * over (X: all of them; X^T: [blk0, blk1) resp. [b0, b1)); inputs are taken to be at level max_level (the cache holds max_level moduli) */
int sfg_matmul_resident_range_rc_dev(sfg_ctx *ctx, const double *cache_dev, int s, int max_level, const sfg_geno *g, unsigned flags,
                                     int blk0, int blk1, uint64_t *out_dev);
int sfg_matmul_accumulate_rc_dev(sfg_ctx *ctx, const double *cache_dev, int s, int max_level, const sfg_geno *g, unsigned flags,
                                 int b0, int b1, int j0, int j1, int accumulate, uint64_t *acc_dev);
/* after an integer all-reduce(sum) of partial outputs across ranks: canonical reduction mod q_l of [rows][L][N] */
int sfg_reduce_rows_dev(sfg_ctx *ctx, uint64_t *rows_dev, size_t nrows_of_L, int L);

/* ---- one party's products on the G GPUs of a node (mgpu.hip) ----
 * The reference runs one OS process per party (run_example.sh:1-12) and calls MatMult4StreamCompute twice per power iteration (gwas/pca.go:344,352) and
 * MatMult4Stream once per SNP batch (gwas/assoc.go:360-408).  sfg_mgpu_create gives such a process all of a node's GPUs: one context per device, one host thread per
 * device inside every call, RCCL (resolved with dlopen at the first use) for the one exchange step of Q' * X^T.  Launchers that start one process per GPU
 * (bench.py under torch.distributed.run) join the same engine with sfg_mgpu_create_rank and a 128-byte id made by sfg_mgpu_unique_id on rank 0 and carried to
 * the other ranks by the caller's own channel (the Go network layer, a TCP store).
 * Partitioning: X (n_ind x m_snp) by blocks of 8192 SNP columns, rank r owns blocks [nblk r / world, nblk (r + 1) / world) (sfg_mgpu_shard).
 *   Q * X     every rank multiplies its own output block columns; no data-path collective.
 *   Q' * X^T  contraction over the rank's blocks; per output block column the canonical uint64 accumulators are REDUCE-SCATTERED over the giant-step axis
 *             (padded to world * ceil(91 / world) slots) on a second queue while the next column is multiplied; every rank reduces mod q and aligns its own giant
 *             steps (key switching is not bit-linear: partial sums must be combined BEFORE the rotations of matmult.go:1474-1494), the aligned partial
 *             outputs are ALL-REDUCED and reduced mod q.  Identical words for every world size (tests/test_gpu_mgpu.py, bench.py's digests).
 * SFG_MGPU_TRANSPORT=direct (single process only; forced when `devices` repeats a device, which RCCL refuses): a rank sums its slice straight from its peers'
 * buffers with a kernel (peer access); SFG_MGPU_FORCE_COLLECTIVES=1 runs the exchange even at world size 1; SFG_MGPU_CACHE_GB (72) bounds a rank's own
 * rotation cache for the pipelined form.  A failing rank fails the call (sfg_mgpu_last_error names it). */
typedef struct sfg_mgpu sfg_mgpu;
typedef struct sfg_mgeno sfg_mgeno;
#define SFG_MGPU_ID_BYTES 128
int sfg_mgpu_unique_id(uint8_t *id128);
/* context arguments as sfg_ctx_create; devices[n]: HIP device indices, world size = n, local rank i = rank i */
int sfg_mgpu_create(sfg_mgpu **out, const int *devices, int n, int logN, int nq, int np, const uint64_t *moduli, const uint64_t *psi, double scale);
int sfg_mgpu_create_rank(sfg_mgpu **out, int device, int rank, int world, const uint8_t *id128, int logN, int nq, int np, const uint64_t *moduli,
                         const uint64_t *psi, double scale);
/* the same with a configuration (sfg_config above; NULL = defaults): every rank's context takes it, the engine takes mgpu_transport / mgpu_cache_bytes / rccl_lib */
int sfg_mgpu_create_ex(sfg_mgpu **out, const int *devices, int n, int logN, int nq, int np, const uint64_t *moduli, const uint64_t *psi, double scale,
                       const sfg_config *config);
int sfg_mgpu_create_rank_ex(sfg_mgpu **out, int device, int rank, int world, const uint8_t *id128, int logN, int nq, int np, const uint64_t *moduli,
                            const uint64_t *psi, double scale, const sfg_config *config);
void sfg_mgpu_destroy(sfg_mgpu *mg);
const char *sfg_mgpu_last_error(const sfg_mgpu *mg);      /* mg may be NULL: error of a failed create */
int sfg_mgpu_world(const sfg_mgpu *mg);
int sfg_mgpu_nlocal(const sfg_mgpu *mg);                  /* ranks driven by this process (n, or 1) */
int sfg_mgpu_rank(const sfg_mgpu *mg, int local);
sfg_ctx *sfg_mgpu_ctx(sfg_mgpu *mg, int local);           /* the context of a local rank: device buffers (sfg_malloc), evaluator ops, phase timers */
const char *sfg_mgpu_transport(const sfg_mgpu *mg);       /* "none" (world 1), "rccl", "direct" */
int sfg_mgpu_synchronize(sfg_mgpu *mg);
/* What the RCCL communicator of local rank `local` itself reports (ncclCommCount / ncclCommUserRank); 0 / 0 when the engine holds none (world 1, direct transport).
 * A record of a multi-GPU run can state "RCCL saw N ranks" from this instead of trusting the launcher's environment. */
int sfg_mgpu_comm_info(sfg_mgpu *mg, int local, int *nranks, int *rank);
/* Pre-flight of the exchange paths, for the first contact with a node (the exchanges stand behind gwas/pca.go:344,352): a reduce-scatter and an all-reduce of
 * a known uint64 pattern (count_per_rank words per rank slice) through the very functions the products use, on the collectives' queue of every rank, checked on the
 * host.  Every exchanging call - this one, sfg_mgpu_matmul* with SFG_TRANSPOSE - has an AGREEMENT POINT after its allocations and before its first collective: a rank
 * that failed so far fails the call on every rank instead of leaving the others waiting in a collective; a rank that fails after it aborts its communicator
 * (ncclCommAbort), which releases the peers, and the engine then refuses further exchanges. */
int sfg_mgpu_preflight(sfg_mgpu *mg, size_t count_per_rank);
/* key material to every local device: cryptoParams.RotKs / Rlk as sfg_ctx_load_rotkey / _relinkey */
int sfg_mgpu_load_rotkey(sfg_mgpu *mg, uint64_t galois_el, const uint64_t *key_host, int montgomery_form);
int sfg_mgpu_load_relinkey(sfg_mgpu *mg, const uint64_t *key_host, int montgomery_form);
int sfg_mgpu_fill_rotkeys_synthetic(sfg_mgpu *mg, const int *rot_left, int nrot, uint64_t seed);
/* SNP-block shard of `rank`: block columns [*blk0, *blk1), columns [*col0, *col1) (any pointer may be NULL) */
int sfg_mgpu_shard(int world, size_t ncol, int rank, size_t *blk0, size_t *blk1, size_t *col0, size_t *col1);
/* MatMult4StreamPreprocess on G GPUs (matmult.go:914-1041): the party's whole row-major int8 matrix (row stride ld); every local rank keeps its column window
 * resident.  _adopt: per-rank windows the caller made on the ranks' contexts (sfg_geno_from_bed / _from_pgen / _from_device; NULL for an empty window), ownership
 * passes.  _synthetic: the window of ONE global synthetic matrix (sfg_fill_geno_window_dev), optionally 2-bit packed - bench.py and the tests. */
int sfg_mgpu_geno_upload(sfg_mgpu *mg, const int8_t *geno_host, size_t nrow, size_t ncol, size_t ld, sfg_mgeno **out);
int sfg_mgpu_geno_adopt(sfg_mgpu *mg, size_t nrow, size_t ncol, sfg_geno *const *shards, sfg_mgeno **out);
int sfg_mgpu_geno_synthetic(sfg_mgpu *mg, size_t nrow, size_t ncol, uint64_t seed, int packed, sfg_mgeno **out);
void sfg_mgpu_geno_free(sfg_mgpu *mg, sfg_mgeno *g);
const sfg_geno *sfg_mgpu_geno_shard(const sfg_mgeno *g, int local);
int sfg_mgpu_geno_dims(const sfg_mgeno *g, size_t *nrow, size_t *ncol);
int sfg_mgpu_geno_blocks(const sfg_mgeno *g, int local, size_t *blk0, size_t *blk1);
int sfg_mgpu_geno_set_plaintext_cache(sfg_mgpu *mg, const sfg_mgeno *g, size_t max_bytes_per_rank);
/* The row-streamed forms of sfg_geno_create / _write_rows / _compare_rows on the sharded matrix (MatMult4StreamPreprocess, gwas/matmult.go:914-1041, reads one row
 * at a time: gwas/filestream.go:414-426): a chunk of whole-matrix rows is scattered to the ranks' column windows; a chunk of the rows of the TRANSPOSE is compared by
 * the ranks whose windows hold those columns.  *ndiff accumulates over the local ranks. */
int sfg_mgpu_geno_create(sfg_mgpu *mg, size_t nrow, size_t ncol, sfg_mgeno **out);
int sfg_mgpu_geno_write_rows(sfg_mgpu *mg, sfg_mgeno *g, size_t row0, size_t nrows, const int8_t *rows_host, size_t ld);
int sfg_mgpu_geno_compare_rows(sfg_mgpu *mg, const sfg_mgeno *g, unsigned flags, size_t row0, size_t nrows, const int8_t *rows_host, size_t ld, uint64_t *ndiff);
/* MatMult4StreamCompute (matmult.go:1043-1236) on the sharded matrix.  Device-pointer form, A_dev[i] / out_dev[i] on local rank i's device:
 *   flags = 0 or SFG_SQUARE  (Q * X):    A_dev[i] = the whole input grid [s][ceil(nrow / 8192)] (replicated);  out_dev[i] = [s][blk1 - blk0], the rank's block columns
 *   | SFG_TRANSPOSE          (Q' * X^T): A_dev[i] = [s][blk1 - blk0], the inputs of the rank's SNP blocks;    out_dev[i] = the whole [s][ceil(nrow / 8192)], on every rank
 * ciphertext layouts as sfg_matmul_resident_dev; stream-ordered on each rank's queue (sfg_mgpu_synchronize waits).
 * Host-pointer form (the Go shim's): Q * X takes A_host [s][nbr] and fills out_host [s][m_ct] (in a multi-process world: the block columns of this process's
 * ranks only); Q' * X^T takes A_host [s][m_ct] (all SNP blocks, each rank uploads its own) and fills out_host [s][nbr] completely in every process. */
int sfg_mgpu_matmul_dev(sfg_mgpu *mg, const uint64_t *const *A_dev, int s, int in_level, int max_level, const sfg_mgeno *g, unsigned flags,
                        uint64_t *const *out_dev);
int sfg_mgpu_matmul(sfg_mgpu *mg, const uint64_t *A_host, int s, int in_level, int max_level, const sfg_mgeno *g, unsigned flags, uint64_t *out_host);
/* GenoBlockMult's batch loop (gwas/assoc.go:340-420) on G GPUs: the reference hands the SNP batches of a chromosome file to assoc_num_blocks_parallel workers
 * (:360-408); here batch k goes to rank k % world.  Every rank streams its batches from the file (sfg_assoc_stream_bed / _pgen: reader thread, pinned slots, decode on
 * the device), multiplies them against its own call-wide rotation cache of `mat` and its output ciphertexts land at their positions of
 * out_host [s][out_ct_capacity][2][max_level][N] (a multi-process world: the positions of this process's ranks only).  A_host: mat, [s][ceil(kept samples / 8192)]
 * ciphertexts at in_level.  sum_host / sqsum_host, flags, *out_ct as in the single-GPU calls.  No collective. */
int sfg_mgpu_assoc_stream_bed(sfg_mgpu *mg, const char *bed_path, size_t num_sample, size_t num_snp, const uint8_t *row_filter, const uint8_t *col_filter,
                              size_t batch_snps, const uint64_t *A_host, int s, int in_level, int max_level, unsigned flags,
                              uint64_t *out_host, size_t out_ct_capacity, size_t *out_ct, double *sum_host, double *sqsum_host);
int sfg_mgpu_assoc_stream_pgen(sfg_mgpu *mg, const char *pgen_path, const uint8_t *row_filter, const uint8_t *col_filter, size_t kept_samples,
                               size_t batch_snps, const uint64_t *A_host, int s, int in_level, int max_level, unsigned flags,
                               uint64_t *out_host, size_t out_ct_capacity, size_t *out_ct, double *sum_host, double *sqsum_host);

/* ---- f-1: collective bootstrap, LOCAL work (mpc/mhe.go:222-348: CollectiveBootstrap / CollectiveBootstrapMat) ----
 * Per ciphertext the reference calls lattigo's dckks.RefreshProtocol: GenShares (mhe.go:251,315), aggregates the shares over the network
 * (AggregateRefreshShare*, stays in Go), then Decrypt / Recode / Recrypt (mhe.go:256-258,329-331).  PARITY UNPINNED: restated from the published
 * lattigo v2.1.0 dckks/refresh.go for ciphertext scale == target scale (the fork's target-scale argument is not in the reference tree).
 * Randomness stays with the caller: mask_dev [nct][N][mask_limbs] two's-complement 64-bit limbs (ring.RandInt(bound) recentred, one big
 * integer per coefficient), e0/e1 [nct][N] Gaussian error coefficients, crs [nct][nq][N] the common reference polynomials (NTT domain). */
/* cryptoParams.Sk.Value (crypto.go:44): secret-key shard, rows [nq][N] in the NTT domain; montgomery_form != 0 for lattigo's stored form */
int sfg_ctx_load_secret_key(sfg_ctx *ctx, const uint64_t *sk_host, int montgomery_form);
/* RefreshProtocol.GenShares on nct ciphertexts [nct][2][level+1][N]:
 *   h0 [nct][level+1][N] = NTT(mask + e0) + sk (.) c1        (refSharesDecrypt)
 *   h1 [nct][nq][N]      = -(NTT(mask + e1) + sk (.) crs)     (refSharesRecrypt, at MaxLevel) */
int sfg_refresh_gen_shares_dev(sfg_ctx *ctx, const uint64_t *ct_dev, int nct, int level, const uint64_t *crs_dev, const uint64_t *mask_dev, int mask_limbs,
                               const int32_t *e0_dev, const int32_t *e1_dev, uint64_t *h0_dev, uint64_t *h1_dev);
/* RefreshProtocol.Decrypt + Recode + Recrypt with the AGGREGATED shares: out [nct][2][nq][N] at MaxLevel = nq-1:
 *   c0' = NTT( centred( CRT( INTT(c0 + h0agg) ) ) mod every q_j ) + h1agg,   c1' = crs */
int sfg_refresh_finish_dev(sfg_ctx *ctx, const uint64_t *ct_dev, int nct, int level, const uint64_t *h0agg_dev, const uint64_t *h1agg_dev,
                           const uint64_t *crs_dev, uint64_t *out_dev);

/* The target-scale form, which is what the reference calls: mhe.go:251,315 GenShares(sk, levelStart, nParties, ct, parameters.Scale(), crp, ...) and
 * mhe.go:257,330 Recode(ct, parameters.Scale()), on products whose scale is A.scale * Params.Scale() (matmult.go:44,92 -> :1045).  PARITY UNPINNED:
 * restated from the published lattigo v2.2.0 dckks/refresh.go (the nearest upstream with the targetScale argument):
 *   GenShares: recrypt share from Quo(mask * Int(target_scale), Int(ct_scale)) (big.Int.Quo, truncated towards zero); decrypt share from the mask itself
 *   Recode:    x <- Quo(x * Int(target_scale), Int(ct_scale)) on the centred big integer, before the re-reduction into all nq moduli
 * The caller sets the refreshed ciphertext's scale to target_scale.  mask_limbs <= 8 here. */
int sfg_refresh_gen_shares_scaled_dev(sfg_ctx *ctx, const uint64_t *ct_dev, int nct, int level, double ct_scale, double target_scale, const uint64_t *crs_dev,
                                      const uint64_t *mask_dev, int mask_limbs, const int32_t *e0_dev, const int32_t *e1_dev, uint64_t *h0_dev, uint64_t *h1_dev);
int sfg_refresh_finish_scaled_dev(sfg_ctx *ctx, const uint64_t *ct_dev, int nct, int level, double ct_scale, double target_scale, const uint64_t *h0agg_dev,
                                  const uint64_t *h1agg_dev, const uint64_t *crs_dev, uint64_t *out_dev);

/* f-4 (partial): ring work of MPC.CMatToSS (mpc/ss.go:146-281): the masked decryption share of each ciphertext and NTT(mask), which the Go side
 * turns into the additive share with the fork's DecodeRVec (ss.go:253-262; fork-only encoder API, stays in Go).
 *   h0 [nct][level+1][N] = NTT(mask) + sk (.) c1 + NTT(e0)   (ss.go:222-236)      mask_ntt [nct][level+1][N] = NTT(mask)   (ctMask, ss.go:226) */
int sfg_ckks_to_ss_share_dev(sfg_ctx *ctx, const uint64_t *ct_dev, int nct, int level, const uint64_t *mask_dev, int mask_limbs, const int32_t *e0_dev,
                             uint64_t *h0_dev, uint64_t *mask_ntt_dev);

/* f-4: the share algebra of MPC.SSToCMat that does not need the fork (mpc/ss.go:84-110; EncodeRVecNew, :125, and the encryption stay in Go).
 * Field elements as in the Beaver products.  rand_dev: the ring.RandInt(bound) draws of the caller (< bound), bound_host = Modulus / (4 (nParty - 1)):
 *   mask = rand >= bound / 2 ? rand - bound : rand  (mod p, :90-99);   rm_masked = rm - mask (:101-102, what RevealSymMat then opens)
 * and, on the hub party after the reveal, share = revealed + mask (:104-106). */
int sfg_ss_mask_dev(sfg_ctx *ctx, int limbs, const uint64_t *modulus_host, const uint64_t *bound_host, const uint64_t *rm_dev, const uint64_t *rand_dev,
                    uint64_t *rm_masked_dev, uint64_t *mask_dev, size_t n);
int sfg_ss_hub_share_dev(sfg_ctx *ctx, int limbs, const uint64_t *modulus_host, const uint64_t *revealed_dev, const uint64_t *mask_dev, uint64_t *share_dev, size_t n);

/* ---- B1-B3: Beaver local products (mpc/beavermult.go:94-147) over a prime field of `limbs` 64-bit LE limbs ---- */
int sfg_beaver_elem_dev(sfg_ctx *ctx, int pid, int limbs, const uint64_t *modulus_host,
                        const uint64_t *ar_dev, const uint64_t *am_dev, const uint64_t *br_dev, const uint64_t *bm_dev,
                        uint64_t *out_dev, size_t n);
int sfg_beaver_elem(sfg_ctx *ctx, int pid, int limbs, const uint64_t *modulus_host,
                    const uint64_t *ar_host, const uint64_t *am_host, const uint64_t *br_host, const uint64_t *bm_host,
                    uint64_t *out_host, size_t n);
int sfg_beaver_matmul(sfg_ctx *ctx, int pid, int limbs, const uint64_t *modulus_host,
                      const uint64_t *ar_host, const uint64_t *am_host, const uint64_t *br_host, const uint64_t *bm_host,
                      uint64_t *out_host, int m, int k, int n);

/* ---- P1: count-sketch + moments (gwas/pca.go:152-162) ----
 * sketch[kp][ncol] fp64 (exact integers), xsum/x2sum[ncol] uint64 */
int sfg_sketch(sfg_ctx *ctx, const sfg_geno *g, const int32_t *bucket_host, const int8_t *sgn_host, int kp,
               double *sketch_host, uint64_t *xsum_host, uint64_t *x2sum_host);

/* ---- synthetic data generators used by bench.py / tests (counter-mode splitmix64) ---- */
int sfg_fill_uniform_ct_dev(sfg_ctx *ctx, uint64_t *ct_dev, int nct, int level, uint64_t seed);
int sfg_fill_geno_dev(sfg_ctx *ctx, int8_t *geno_dev, size_t nrow, size_t ncol, uint64_t seed);
/* the column window [col0, col0+ncol) of the same global nrow x ncol_global matrix (row stride ld): SNP-sharded ranks hold
 * exactly the bytes the single-GPU run holds in that window, so outputs are comparable across world sizes */
int sfg_fill_geno_window_dev(sfg_ctx *ctx, int8_t *geno_dev, size_t nrow, size_t ncol, size_t ld, size_t col0, size_t ncol_global, uint64_t seed);
int sfg_fill_rotkeys_synthetic(sfg_ctx *ctx, const int *rot_left, int nrot, uint64_t seed);

/* last kernel timing hooks for bench.py: milliseconds spent (HIP events on the ctx stream) in the named phase
 * of the most recent matmul call: "encode" (diagonal FFT + plaintext NTT), "mac", "mac_small"/"mac_big" (the two k_mac instances), "rotate", "skew" ; returns <0 if unknown */
int sfg_ctx_clear_phases(sfg_ctx *ctx);
double sfg_last_phase_ms(const sfg_ctx *ctx, const char *phase);
int sfg_last_phase_launches(const sfg_ctx *ctx, const char *phase);
/* algorithmic bytes (operands read once + results written once, as laid out in HBM) credited to the phase's launches */
double sfg_last_phase_bytes(const sfg_ctx *ctx, const char *phase);

#ifdef __cplusplus
}
#endif
#endif
