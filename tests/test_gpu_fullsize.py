"""GPU-vs-oracle parity at the launch shapes the small-matrix tests never reach:

  * one FULL 8192 x 8192 block: all 8192 diagonals, 91 x 91 plaintext panel, half-row + mirrored panels;
  * several block rows fused into one MAC launch (K = G*91), accumulation across groups, two column passes with the
    giant-step alignment overlapped on the auxiliary queue, ragged edge blocks — all against the oracle, every word;
  * the two-phase (contraction-sharded) entry points sfg_matmul_accumulate_dev / sfg_matmul_finalize_dev against
    orc_matmult_accumulate / orc_matmult_finalize on block-row and giant sub-ranges;
  * the BASELINE.json configuration shapes: c1 stand-in (1000 x 100 000, s = 13, computeSquaredSum), c2 (10 000 x 100 000,
    kp = 15, both products), c5 batch (500 000 x 8192, s in {13, 1}, square) — oracle on the block columns / block-row
    ranges it can finish in seconds, size-independent properties for the rest.

The oracle runs on the host cores of the GPU box (OpenMP over giant steps, oracle/sfgwas_oracle.c); the out[i][j] block of
a product depends only on block column j of the matrix and row i of A, so columns / rows are checked independently."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
SLOTS, D, N, L, LEVEL = 8192, 91, 16384, 5, 5
SCALE = 2.0 ** 34


class Env:
    """one GPU context + the oracle ring with the SAME (uniform random) rotation keys"""

    def __init__(self, **env_overrides):
        from sfgwas_amd import capi
        from sfgwas_amd.params import rotations_for_matmul
        self.capi = capi
        saved = {k: os.environ.get(k) for k in env_overrides}
        os.environ.update({k: str(v) for k, v in env_overrides.items()})
        try:
            self.ctx = capi.Context(ol.Q_PN14, ol.P_PN14)        # the library reads its A/B switches here, once
        finally:
            for k, v in saved.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        self.ring = ol.Ring(14, ol.Q_PN14, ol.P_PN14)
        self.keys = ol.RotKeys(self.ring)
        for k in rotations_for_matmul():
            g = self.ring.galois(k)
            key = capi.random_rotkey(self.ring.moduli, self.ring.beta, self.ring.N, 1000 + k)
            self.keys.add(g, key)
            self.ctx.load_rotkey(g, key)

    def close(self):
        self.ctx.close()


@pytest.fixture(scope="module")
def env():
    e = Env()
    yield e
    e.close()


def host_cts(ring, s, nbr, level, seed):
    return np.stack([np.stack([ring.fill_uniform(level, seed * 1000 + i * 100 + b) for b in range(nbr)]) for i in range(s)])


def oracle_product(env, A, geno_logical, square=False, in_level=LEVEL):
    """orc_matmult4stream, one block column at a time (bounds the oracle's u128 accumulator memory)"""
    s = A.shape[0]
    ncol = geno_logical.shape[1]
    cols = []
    for j in range((ncol - 1) // SLOTS + 1):
        sub = np.ascontiguousarray(geno_logical[:, j * SLOTS:(j + 1) * SLOTS])
        out, _, _ = ol.matmult4stream(env.ring, env.keys, SCALE, A, in_level, L, sub, square=square, enc_prec=1)
        cols.append(out)
    return np.concatenate(cols, axis=1) if len(cols) > 1 else cols[0]


def digest(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


# ------------------------------------------------------------------------------------------- full block, all diagonals
def test_full_block_all_8192_diagonals_vs_oracle(env):
    rnd = np.random.default_rng(101)
    geno = rnd.integers(-1, 3, (SLOTS, SLOTS)).astype(np.int8)
    A = host_cts(env.ring, 1, 1, LEVEL, 3)
    capi = env.capi
    g = env.ctx.geno_upload(geno)
    dA = capi.DevArray.from_host(env.ctx, A)
    got = env.ctx.matmul_resident(dA, 1, LEVEL, L, g)
    want = oracle_product(env, A, geno)
    h = got.host()
    assert np.array_equal(h, want), f"{np.count_nonzero(h != want)} words differ"
    # the transposed read of the same resident copy against the oracle on X^T
    got_t = env.ctx.matmul_resident(dA, 1, LEVEL, L, g, flags=capi.SFG_TRANSPOSE)
    want_t = oracle_product(env, A, np.ascontiguousarray(geno.T))
    assert np.array_equal(got_t.host(), want_t)
    for d in (dA, got, got_t):
        d.free()
    env.ctx.geno_free(g)


# ------------------------------------------------------------- fused groups, accumulate across groups, two column passes
def test_multi_group_two_pass_overlap_vs_oracle():
    """3 block rows x 2 block columns (ragged last row and column), s = 2, with SFG_MM_GROUP = 2 (two MAC launches per
    block column, the second accumulating onto the first) and an accumulator budget of one block column per pass (two column passes).
    (The two-queue schedule this test also exercised until round 5 is an A/B-build switch now; tests/test_gpu_properties.py holds it against the default.)"""
    s, nrow, ncol = 2, 2 * SLOTS + 100, SLOTS + 50
    accw_mb = D * s * 2 * L * N * 8 / 2 ** 20                       # one block column of accumulators
    e = Env(SFG_MM_GROUP=2, SFG_MM_ACC_BUDGET_MB=int(2 * accw_mb) + 1)
    try:
        rnd = np.random.default_rng(202)
        geno = rnd.integers(-1, 3, (nrow, ncol)).astype(np.int8)
        A = host_cts(e.ring, s, 3, LEVEL, 5)
        g = e.ctx.geno_upload(geno)
        dA = e.capi.DevArray.from_host(e.ctx, A)
        got = e.ctx.matmul_resident(dA, s, LEVEL, L, g)
        want = oracle_product(e, A, geno)
        h = got.host()
        assert np.array_equal(h, want), f"{np.count_nonzero(h != want)} words differ"
        dA.free(); got.free(); e.ctx.geno_free(g)
    finally:
        e.close()


# ------------------------------------------------------------------------------------ two-phase entry points vs oracle
def oracle_accumulate(env, A, geno_logical, b0, b1, square=False):
    s, nbr = A.shape[0], A.shape[1]
    nrow, ncol = geno_logical.shape
    m_ct = (ncol - 1) // SLOTS + 1
    acc = np.zeros((m_ct, D, s, 2, L, N), dtype=np.uint64)
    ga = np.zeros(D, dtype=np.uint8)
    rc = ol.lib().orc_matmult_accumulate(env.ring.h, env.keys.h, SCALE, ol.p64(np.ascontiguousarray(A)), s, LEVEL, L,
                                         ol.pi8(np.ascontiguousarray(geno_logical)), nrow, ncol, int(square), 1, b0, b1, ol.p64(acc),
                                         ga.ctypes.data_as(C.POINTER(C.c_uint8)))
    assert rc == 0
    return acc, ga


def oracle_finalize(env, acc, s, m_ct, g0, g1, out=None):
    accumulate = out is not None
    if out is None:
        out = np.zeros((s, m_ct, 2, L, N), dtype=np.uint64)
    rc = ol.lib().orc_matmult_finalize(env.ring.h, env.keys.h, L, s, m_ct, ol.p64(np.ascontiguousarray(acc)), None, g0, g1, int(accumulate), ol.p64(out))
    assert rc == 0
    return out


def test_two_phase_accumulate_finalize_vs_oracle(env):
    """contraction-sharded form: X^T operand with 3 block rows (2 x 8192 + 300 SNPs) and 2 block columns of individuals;
    per 'rank' accumulators over block-row sub-ranges and giant-step sub-ranges of the alignment, each against the oracle"""
    rnd = np.random.default_rng(303)
    s = 2
    X = rnd.integers(-1, 3, (SLOTS + 40, 2 * SLOTS + 300)).astype(np.int8)     # stored n_ind x m_snp; operand = X^T
    Xt = np.ascontiguousarray(X.T)
    nbr, m_ct = 3, 2
    A = host_cts(env.ring, s, nbr, LEVEL, 9)
    capi = env.capi
    g = env.ctx.geno_upload(X)
    dA = capi.DevArray.from_host(env.ctx, A)
    parts = []
    for (b0, b1) in [(0, 1), (1, 3)]:
        acc = env.ctx.matmul_accumulate(dA, s, LEVEL, L, g, capi.SFG_TRANSPOSE, b0, b1, 0, m_ct)
        want, _ = oracle_accumulate(env, A, Xt, b0, b1)
        h = acc.host()
        assert np.array_equal(h, want), f"accumulate [{b0},{b1}): {np.count_nonzero(h != want)} words differ"
        parts.append(h)
        acc.free()
    tot = parts[0] + parts[1]                                                   # partial accumulators add up mod q (< 2^47: no overflow)
    for l in range(L):
        tot[..., l, :] %= np.uint64(ol.Q_PN14[l])
    # a block-COLUMN sub-range of the accumulate over all block rows equals that slice of the summed parts
    accj = env.ctx.matmul_accumulate(dA, s, LEVEL, L, g, capi.SFG_TRANSPOSE, 0, nbr, 1, 2)
    assert np.array_equal(accj.host()[0], tot[1])
    accj.free()
    dtot = capi.DevArray.from_host(env.ctx, tot)
    out = None
    want_out = None
    for (g0, g1) in [(0, 30), (30, 91)]:                                       # giant-step shards, accumulated
        out = env.ctx.matmul_finalize(dtot, s, L, m_ct, g0, g1, out=out)
        want_out = oracle_finalize(env, tot, s, m_ct, g0, g1, out=want_out)
        assert np.array_equal(out.host(), want_out), f"finalize giants [{g0},{g1})"
    one_shot = env.ctx.matmul_resident(dA, s, LEVEL, L, g, flags=capi.SFG_TRANSPOSE)
    assert np.array_equal(one_shot.host(), want_out)
    # what rank 7 of 8 holds after the reduce-scatter over the padded giant axis: slots 84..95 (84..90 real) of every block column
    gpr, base = 12, 84
    chunk = np.zeros((m_ct, gpr, s, 2, L, N), dtype=np.uint64)
    chunk[:, :D - base] = tot[:, base:]
    chunk[:, D - base:] = 12345                                                # slots past giant 90 must be ignored
    dchunk = capi.DevArray.from_host(env.ctx, chunk)
    part = capi.DevArray(env.ctx, (s, m_ct, 2, L, N))
    env.ctx.check(capi.lib().sfg_matmul_finalize_slots_dev(env.ctx.h, dchunk.p, s, L, m_ct, gpr, base, 0, gpr, 0, part.p), "finalize_slots")
    assert np.array_equal(part.host(), oracle_finalize(env, tot, s, m_ct, base, D))
    dchunk.free(); part.free()
    for d in (dA, dtot, out, one_shot):
        d.free()
    env.ctx.geno_free(g)


# --------------------------------------------------------------------------------------------- BASELINE.json config shapes
def test_c1_standin_1000x100000_s13_sums_vs_oracle(env):
    """configs[0] stand-in: example_data-sized 1000 x 100 000 per party through MatMult4Stream with s = 13
    (ncov + 1 + npc + 2, assoc.go:699-704) and computeSquaredSum — the host-pointer entry point, as assoc.go:424 calls it"""
    nrow, ncol, s = 1000, 100_000, 13
    rnd = np.random.default_rng(404)
    maf = rnd.uniform(0.05, 0.5, ncol)
    geno = rnd.binomial(2, maf, (nrow, ncol)).astype(np.int8)
    geno[rnd.random((nrow, ncol)) < 0.01] = -1
    A = host_cts(env.ring, s, 1, LEVEL, 11)
    got, sm, sq = env.ctx.matmul_stream(A, LEVEL, L, geno, want_sums=True)
    clean = np.where(geno < 0, 0, geno).astype(np.float64)
    assert np.array_equal(sm, clean.sum(0)) and np.array_equal(sq, (clean * clean).sum(0))
    m_ct = (ncol - 1) // SLOTS + 1
    assert got.shape == (s, m_ct, 2, L, N)
    for j in (0, 6, m_ct - 1):                                                  # first, middle, ragged last block column
        sub = np.ascontiguousarray(geno[:, j * SLOTS:(j + 1) * SLOTS])
        want, _, _ = ol.matmult4stream(env.ring, env.keys, SCALE, A, LEVEL, L, sub, enc_prec=1)
        assert np.array_equal(got[:, j], want[:, 0]), f"block column {j}"
    # row independence: the s = 1 call on row 4 of A reproduces row 4 of the s = 13 result in every block column
    got1, _, _ = env.ctx.matmul_stream(A[4:5], LEVEL, L, geno)
    assert np.array_equal(got1[0], got[4])


def test_c2_10000x100000_kp15_both_products(env):
    """configs[1]: 10 000 x 100 000, kp = 15, Q*X and Q'*X^T on one resident copy.  Oracle: the ragged last block column of Q*X at
    kp = 15 (a full one at kp = 15 is held against the oracle by the c4 test below and the full-block test above) and, for Q'*X^T, rows 0 and 14 of the
    output block column of the last 1808 individuals; the remaining words are tied to those by the range/row properties below."""
    n_ind, m_snp, kp = 10_000, 100_000, 15
    capi = env.capi
    gd, g = env.ctx.fill_geno(n_ind, m_snp, 0x5F6A + 2)
    geno = gd.host()
    nbr_x, mct_x = 2, 13
    A1 = env.ctx.fill_uniform_cts(kp * nbr_x, LEVEL, 0xC1F3)
    A2 = env.ctx.fill_uniform_cts(kp * mct_x, LEVEL, 0xC1F4)
    out1 = env.ctx.matmul_resident(A1, kp, LEVEL, L, g)                         # [kp][13]
    out2 = env.ctx.matmul_resident(A2, kp, LEVEL, L, g, flags=capi.SFG_TRANSPOSE)   # [kp][2]
    h1 = out1.host().reshape(kp, mct_x, 2, L, N)
    h2 = out2.host().reshape(kp, nbr_x, 2, L, N)
    A1h = A1.host().reshape(kp, nbr_x, 2, LEVEL + 1, N)
    for j in (mct_x - 1,):
        sub = np.ascontiguousarray(geno[:, j * SLOTS:(j + 1) * SLOTS])
        want, _, _ = ol.matmult4stream(env.ring, env.keys, SCALE, A1h, LEVEL, L, sub, enc_prec=1)
        assert np.array_equal(h1[:, j], want[:, 0]), f"Q*X block column {j}"
    # Q'*X^T: accumulators of contraction block rows [3, 5) x output block column 1 (the last 1808 individuals), rows 0 and 14 of A
    A2h = A2.host().reshape(kp, mct_x, 2, LEVEL + 1, N)
    sub_t = np.ascontiguousarray(geno[SLOTS:, 3 * SLOTS:5 * SLOTS].T)           # X^T block rows 3..4, block column 1
    rows = [0, 14]
    dsel = capi.DevArray.from_host(env.ctx, np.ascontiguousarray(A2h[rows]))
    acc_sel = env.ctx.matmul_accumulate(dsel, 2, LEVEL, L, g, capi.SFG_TRANSPOSE, 3, 5, 1, 2)
    want_acc, _ = oracle_accumulate(env, np.ascontiguousarray(A2h[rows][:, 3:5]), sub_t, 0, 2)
    assert np.array_equal(acc_sel.host(), want_acc), "Q'*X^T accumulators, block rows [3,5), block column 1"
    acc_sel.free(); dsel.free()
    # properties over everything else: output-range concatenation, contraction two-phase, digest stable across a re-run
    parts = [env.ctx.matmul_resident(A1, kp, LEVEL, L, g, blk=(a, b)) for a, b in ((0, 6), (6, 13))]
    assert np.array_equal(np.concatenate([p.host() for p in parts], axis=1).reshape(h1.shape), h1)
    acc = env.ctx.matmul_accumulate(A2, kp, LEVEL, L, g, capi.SFG_TRANSPOSE, 0, mct_x, 0, nbr_x)
    fin = env.ctx.matmul_finalize(acc, kp, L, nbr_x, 0, D)
    assert np.array_equal(fin.host().reshape(h2.shape), h2)
    # giant-step alignment of the FULL kp = 15 accumulators by the oracle: every word of Q'*X^T
    assert np.array_equal(oracle_finalize(env, acc.host(), kp, nbr_x, 0, D), h2), "Q'*X^T finalize vs oracle"
    again = env.ctx.matmul_resident(A2, kp, LEVEL, L, g, flags=capi.SFG_TRANSPOSE)
    assert digest(again.host()) == digest(h2)
    for d in (A1, A2, out1, out2, acc, fin, again, gd, *parts):
        d.free()
    env.ctx.geno_free(g)


def test_c5_batch_500000x8192_square_s13_and_s1(env):
    """configs[4] association batch (assoc.go:371-416, gWY :1338-1422): 500 000 individuals x 8192 SNPs per MatMult4Stream call,
    s = 13 (WzBT-style) and s = 1 with square = true (:1375).  Oracle: the accumulators of block rows [30, 32) (two full
    8192 x 8192 blocks), plain and squared; properties: squared flag == explicitly squared upload, row independence,
    two-phase == one-shot."""
    n_ind, m_snp = 500_000, SLOTS
    capi = env.capi
    nbr = (n_ind - 1) // SLOTS + 1                                              # 62
    gd, g = env.ctx.fill_geno(n_ind, m_snp, 0x5F6A + 5)
    A13 = env.ctx.fill_uniform_cts(13 * nbr, LEVEL, 77)
    out13 = env.ctx.matmul_resident(A13, 13, LEVEL, L, g)                       # (13 x n_ind) * (n_ind x 8192): 62 block rows, 1 block column
    h13 = out13.host()
    assert h13.shape == (13, 1, 2, L, N)
    # s = 1 on row 7 of A: row independence
    A13h_row7 = np.stack([A13.host_slice((7 * nbr + b,)) for b in range(nbr)])[None]       # [1][nbr][2][6][N]
    d7 = capi.DevArray.from_host(env.ctx, A13h_row7)
    out1 = env.ctx.matmul_resident(d7, 1, LEVEL, L, g)
    assert np.array_equal(out1.host()[0], h13[7])
    # square = true (s = 1): flag vs an explicitly squared resident copy
    outsq = env.ctx.matmul_resident(d7, 1, LEVEL, L, g, flags=capi.SFG_SQUARE)
    # oracle on block rows [30, 32) of the contraction: accumulators, plain and squared
    r0, r1 = 30 * SLOTS, 32 * SLOTS
    sub = np.empty((r1 - r0, m_snp), dtype=np.int8)
    for i in range(r0, r1, 4096):
        n = min(4096, r1 - i)
        blk = np.empty((n, m_snp), dtype=np.int8)
        src = C.c_void_p(gd.p.value + i * m_snp)
        env.ctx.check(capi.lib().sfg_memcpy_d2h(env.ctx.h, blk.ctypes.data_as(C.c_void_p), src, blk.nbytes), "d2h")
        sub[i - r0:i - r0 + n] = blk
    Asub = np.ascontiguousarray(A13h_row7[:, 30:32])
    for sq_flag in (0, capi.SFG_SQUARE):
        acc = env.ctx.matmul_accumulate(d7, 1, LEVEL, L, g, sq_flag, 30, 32, 0, 1)
        want, _ = oracle_accumulate(env, Asub, sub, 0, 2, square=bool(sq_flag))
        assert np.array_equal(acc.host(), want), f"accumulate block rows [30,32) square={bool(sq_flag)}"
        acc.free()
    # two-phase over all 62 block rows == one-shot, for the squared product
    acc = env.ctx.matmul_accumulate(d7, 1, LEVEL, L, g, capi.SFG_SQUARE, 0, 31, 0, 1)
    acc = env.ctx.matmul_accumulate(d7, 1, LEVEL, L, g, capi.SFG_SQUARE, 31, nbr, 0, 1, acc=acc)
    fin = env.ctx.matmul_finalize(acc, 1, L, 1, 0, D)
    assert np.array_equal(fin.host(), outsq.host())
    assert np.array_equal(oracle_finalize(env, acc.host(), 1, 1, 0, D), outsq.host()), "finalize vs oracle"
    for d in (A13, out13, d7, out1, outsq, acc, fin, gd):
        d.free()
    env.ctx.geno_free(g)


def test_c5_streamed_bed_batches_on_the_int8_rot_tiles_equal_the_resident_products(env, tmp_path):
    """configs[4] as sfg_assoc_stream_bed runs it (assoc.go:371-416): 500 000 individuals, s = 13, two batches (2048 and 952 kept SNPs) streamed from a .bed.
    The call keeps the baby-step rotation cache of all 62 block rows as the int8 MAC's rot tiles (round 4: 8 MAC groups of 8, 8, ..., 6 block rows, the last
    block row ragged) and multiplies every batch from them; each batch must equal, word for word, MatMult4Stream of the same ciphertexts with the decoded batch
    resident - the product path the test above holds against the oracle (its own rotations, transposed per launch)."""
    capi = env.capi
    n_ind, nsnp, batch, s = 500_000, 3000, 2048, 13
    bps = n_ind // 4
    rnd = np.random.default_rng(0xBED5)
    raw = rnd.integers(0, 256, (nsnp, bps), dtype=np.uint8)
    path = str(tmp_path / "c5.bed")
    with open(path, "wb") as f:
        f.write(bytes([0x6C, 0x1B, 0x01])); f.write(raw.tobytes())
    nbr = (n_ind - 1) // SLOTS + 1
    env.ctx.check(capi.lib().sfg_ctx_release_scratch(env.ctx.h), "release_scratch")
    A = env.ctx.fill_uniform_cts(s * nbr, LEVEL, 0xA550C)
    out = capi.DevArray(env.ctx, (s, 2, 2, L, N))
    got = C.c_size_t()
    env.ctx.check(capi.lib().sfg_assoc_stream_bed(env.ctx.h, path.encode(), n_ind, nsnp, None, None, batch, A.p, s, LEVEL, L, 0, out.p, 2, C.byref(got), None, None), "assoc_stream_bed")
    assert got.value == 2
    kept = C.c_size_t()
    env.ctx.check(capi.lib().sfg_ctx_scratch_bytes(env.ctx.h, b"assoc.rot8", C.byref(kept)), "scratch_bytes")
    assert kept.value > 60 << 30, "the scan did not keep its rotation cache as int8 tiles"
    h = out.host()
    lut = np.array([2, -1, 1, 0], dtype=np.int8)                                  # 2-bit code -> dosage (scripts/plinkBedToBinary.py:18-27)
    for b, (a0, a1) in enumerate([(0, batch), (batch, nsnp)]):
        geno = np.empty((n_ind, a1 - a0), dtype=np.int8)
        for k in range(4):
            geno[k::4] = lut[(raw[a0:a1] >> (2 * k)) & 3].T
        g = env.ctx.geno_upload(geno)
        want = env.ctx.matmul_resident(A, s, LEVEL, L, g)
        assert np.array_equal(h[:, b:b + 1], want.host()), f"batch {b}: streamed product on the int8 rot tiles differs from the resident product"
        want.free(); env.ctx.geno_free(g)
    A.free(); out.free()
    # the scan's cache, panels and staging stay in the context's pools (> 120 GB here) for its next call; a caller's own allocation that no longer fits beside
    # them gets them back (sfg_malloc retries once after sfg_ctx_release_scratch) instead of failing
    env.ctx.check(capi.lib().sfg_ctx_scratch_bytes(env.ctx.h, b"", C.byref(kept)), "scratch_bytes")
    assert kept.value > 120 << 30
    big = C.c_void_p()
    env.ctx.check(capi.lib().sfg_malloc(env.ctx.h, C.byref(big), C.c_size_t(200 << 30)), "a 200 GB buffer beside the kept pools")
    env.ctx.check(capi.lib().sfg_ctx_scratch_bytes(env.ctx.h, b"", C.byref(kept)), "scratch_bytes")
    assert kept.value == 0
    env.ctx.check(capi.lib().sfg_free(env.ctx.h, big), "free")


# --------------------------------------------------------------------------- configs[3]: 100 000 x 1 000 000 on one GPU
# Digests of bench.py's c4 outputs (seed 0x5F6A genotypes, 0xC1F3 / 0xD2A7_0000 ciphertexts, 0xBEEF keys): identical in BENCH_r02.json (fp64 MAC,
# 8 block rows per launch), BENCH_r03.json (int8 matrix-core MAC, memory-chosen groups) and every profiles/r0*_bench_c4_* line.
C4_OUT1_SHA256 = "cab05b5a8326ff9dc51f0e139c2261d47dc2a8830541a134273bffbc6f3888e6"
C4_OUT2_SHA256 = "ce9b0e28cb6318dafb77b27fbdff1d4491f3fcb3e70548d5bfac723753d7ee62"


def _ct_digest(h):
    """bench.py's digest: SHA-256 over the per-ciphertext SHA-256s in [i][j] order"""
    return hashlib.sha256(b"".join(hashlib.sha256(h[i, j].tobytes()).digest() for i in range(h.shape[0]) for j in range(h.shape[1]))).hexdigest()


def test_c4_100000x1000000_kp15_pinned_digests_and_oracle_at_the_c4_launch_shapes(env):
    """configs[3] (the configuration BASELINE.json's metric is quoted on) at full size, as bench.py runs it: 123 x 13 blocks, kp = 15, the default
    schedule (memory-chosen MAC groups beside 100 GB of int8 genotypes, several accumulator passes).
      (a) both products' digests equal the values pinned since round 2;
      (b) Q*X (matmult.go:1043-1236 via pca.go:344): block columns 61 (interior) and 122 (ragged: 576 SNPs) x rows {0, 14} of the FULL run vs the
          oracle over all 13 block rows (row independence: the oracle runs s = 2);
      (c) Q'*X^T (pca.go:352): the device accumulators of the 14-block-row range [109, 123) (one MAC launch of K = 1274, ragged last SNP block) x the
          ragged output block column 12, rows {0, 14}, vs orc_matmult_accumulate; the device accumulators of ALL 123 block rows for those two rows,
          aligned by orc_matmult_finalize, must equal rows {0, 14} of the full run's output (every block column)."""
    from sfgwas_amd import capi
    from sfgwas_amd.params import rotations_for_matmul
    import time
    t_start = time.perf_counter()
    n_ind, m_snp, kp = 100_000, 1_000_000, 15
    nbr_x, mct_x = 13, 123
    lib = capi.lib()
    env.ctx.check(lib.sfg_ctx_release_scratch(env.ctx.h), "release_scratch")      # the module's context keeps the pools of the c2 / c5 tests: this test needs the whole device
    ctx = capi.Context(ol.Q_PN14, ol.P_PN14)
    ring = ol.Ring(14, ol.Q_PN14, ol.P_PN14)
    keys = ol.RotKeys(ring)
    rots = rotations_for_matmul()
    arr = (C.c_int * len(rots))(*rots)
    ctx.check(lib.sfg_fill_rotkeys_synthetic(ctx.h, arr, len(rots), 0xBEEF), "fill rotkeys")
    for k in rots:
        g_el = ring.galois(k)
        keys.add(g_el, ctx.export_rotkey(g_el))

    class E:                                    # what oracle_product / oracle_accumulate / oracle_finalize read
        pass
    e = E()
    e.ring, e.keys = ring, keys
    try:
        geno = capi.DevArray(ctx, (n_ind, m_snp), np.int8)
        ctx.check(lib.sfg_fill_geno_window_dev(ctx.h, geno.p, n_ind, m_snp, m_snp, 0, m_snp, 0x5F6A), "fill_geno")
        g = C.c_void_p()
        ctx.check(lib.sfg_geno_from_device(ctx.h, geno.p, n_ind, m_snp, m_snp, C.byref(g)), "geno_from_device")
        A1 = ctx.fill_uniform_cts(kp * nbr_x, LEVEL, 0xC1F3)                        # [kp][13]
        A2 = capi.DevArray(ctx, (kp, mct_x, 2, LEVEL + 1, N))                       # [kp][123]: ciphertext (i, b) has seed base + i*123 + b
        ctw = 2 * (LEVEL + 1) * N
        for i in range(kp):
            ctx.check(lib.sfg_fill_uniform_ct_dev(ctx.h, C.c_void_p(A2.p.value + i * mct_x * ctw * 8), mct_x, LEVEL, 0xD2A7_0000 + i * mct_x), "fill A2")
        out1 = ctx.matmul_resident(A1, kp, LEVEL, L, g)                             # Q*X    [kp][123]
        out2 = ctx.matmul_resident(A2, kp, LEVEL, L, g, flags=capi.SFG_TRANSPOSE)   # Q'*X^T [kp][13]
        h2 = out2.host().reshape(kp, nbr_x, 2, L, N)
        assert _ct_digest(h2) == C4_OUT2_SHA256, "Q'*X^T digest moved"
        rows = [0, 14]
        h1_rows = {}
        d1 = hashlib.sha256()
        for i in range(kp):                                                          # 2.4 GB: one row of A at a time
            hi = out1.host_slice((i,)).reshape(mct_x, 2, L, N)
            d1.update(b"".join(hashlib.sha256(hi[j].tobytes()).digest() for j in range(mct_x)))
            if i in rows:
                h1_rows[i] = hi[[61, 122]].copy()
        assert d1.hexdigest() == C4_OUT1_SHA256, "Q*X digest moved"
        out1.free(); out2.free()
        ctx.check(lib.sfg_ctx_release_scratch(ctx.h), "release_scratch")           # the kp = 15 pools; the s = 2 samples below choose their own (larger) groups
        t_products = time.perf_counter()

        def window(c0, ncol, r0=0, r1=n_ind):
            """host copy of X[r0:r1, c0:c0+ncol], regenerated on the device as the window of the same global matrix"""
            w = capi.DevArray(ctx, (n_ind, ncol), np.int8)
            ctx.check(lib.sfg_fill_geno_window_dev(ctx.h, w.p, n_ind, ncol, ncol, c0, m_snp, 0x5F6A), "window")
            out = np.empty((r1 - r0, ncol), dtype=np.int8)
            ctx.check(lib.sfg_memcpy_d2h(ctx.h, out.ctypes.data_as(C.c_void_p), C.c_void_p(w.p.value + r0 * ncol), out.nbytes), "d2h")
            w.free()
            return out

        # (b) Q*X, two block columns x two rows of A over all 13 block rows
        A1h = A1.host().reshape(kp, nbr_x, 2, LEVEL + 1, N)
        A1sel = np.ascontiguousarray(A1h[rows])
        for jj, j in enumerate((61, 122)):
            ncol = min(SLOTS, m_snp - j * SLOTS)
            sub = window(j * SLOTS, ncol)
            # spot check that the window IS the resident matrix there
            probe = np.empty(ncol, dtype=np.int8)
            ctx.check(lib.sfg_memcpy_d2h(ctx.h, probe.ctypes.data_as(C.c_void_p), C.c_void_p(geno.p.value + 77_777 * m_snp + j * SLOTS), ncol), "d2h")
            assert np.array_equal(probe, sub[77_777])
            want, _, _ = ol.matmult4stream(ring, keys, SCALE, A1sel, LEVEL, L, sub, enc_prec=1)
            for r, i in enumerate(rows):
                assert np.array_equal(h1_rows[i][jj], want[r, 0]), f"Q*X row {i} block column {j}"
        t_qx = time.perf_counter()

        # (c) Q'*X^T: accumulators of block rows [109, 123) x output block column 12 (the last 1696 individuals)
        A2sel = np.ascontiguousarray(np.stack([A2.host_slice((i,)) for i in rows]))   # [2][123][2][6][N]
        dsel = capi.DevArray.from_host(ctx, A2sel)
        b0, b1, jo = 109, 123, 12
        sub_t = np.ascontiguousarray(window(b0 * SLOTS, m_snp - b0 * SLOTS, jo * SLOTS, n_ind).T)     # X^T block rows 109..122 x individuals 98304..
        acc_sel = ctx.matmul_accumulate(dsel, 2, LEVEL, L, g, capi.SFG_TRANSPOSE, b0, b1, jo, jo + 1)
        want_acc, _ = oracle_accumulate(e, np.ascontiguousarray(A2sel[:, b0:b1]), sub_t, 0, b1 - b0)
        assert np.array_equal(acc_sel.host(), want_acc), "Q'*X^T accumulators, block rows [109,123), block column 12"
        acc_sel.free()
        t_acc = time.perf_counter()
        acc = ctx.matmul_accumulate(dsel, 2, LEVEL, L, g, capi.SFG_TRANSPOSE, 0, mct_x, 0, nbr_x)
        fin = oracle_finalize(e, acc.host(), 2, nbr_x, 0, D)
        for r, i in enumerate(rows):
            assert np.array_equal(fin[r], h2[i]), f"Q'*X^T row {i}: oracle alignment of the device accumulators vs the full run"
        acc.free(); dsel.free()
        print(f"\nc4 test: products + digests {t_products - t_start:.0f} s, Q*X oracle {t_qx - t_products:.0f} s, accumulate oracle {t_acc - t_qx:.0f} s, "
              f"finalize oracle {time.perf_counter() - t_acc:.0f} s")
        A1.free(); A2.free()
        lib.sfg_geno_free(ctx.h, g)
        geno.free()
    finally:
        ctx.close()
