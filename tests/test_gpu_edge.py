"""Edge cases of the C-ABI on the GPU: degenerate sizes, all-missing / all-zero genotypes, zero-length batches and the error paths the
reference turns into log.Fatal (missing rotation key, level out of range).  Results that exist are compared with the oracle, every word."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
SLOTS, D = 8192, 91


@pytest.fixture(scope="module")
def env():
    from sfgwas_amd import capi
    ctx = capi.Context(ol.Q_PN14, ol.P_PN14)
    ring = ol.Ring(14, ol.Q_PN14, ol.P_PN14)
    keys = ol.RotKeys(ring)
    yield ctx, ring, keys
    ctx.close()


def load_keys(ctx, ring, keys, rots_left):
    from sfgwas_amd import capi
    for k in rots_left:
        g = ring.galois(k)
        if g not in keys.keys:
            key = capi.random_rotkey(ring.moduli, ring.beta, ring.N, 4000 + k)
            keys.add(g, key)
            ctx.load_rotkey(g, key)


def active_steps(nrow, ncol):
    """left rotations MatMult4Stream touches for an nrow x ncol operand (matmult.go:1329-1336): baby steps and d * giant steps"""
    babies, giants = set(), set()
    for bi in range((nrow - 1) // SLOTS + 1):
        nr = min((bi + 1) * SLOTS, nrow) - bi * SLOTS
        for bj in range((ncol - 1) // SLOTS + 1):
            nc = min((bj + 1) * SLOTS, ncol) - bj * SLOTS
            shifts = set(range(0, nr)) | set(range(SLOTS - nc + 1, SLOTS)) if nr + nc <= SLOTS else set(range(SLOTS))
            for sh in shifts:
                babies.add(sh % D); giants.add(sh // D)
    return [b for b in sorted(babies) if b] + [g * D for g in sorted(giants) if g]


def product(env, geno, s, seed, flags=0):
    ctx, ring, keys = env
    load_keys(ctx, ring, keys, active_steps(*geno.shape))
    nbr = (geno.shape[0] - 1) // SLOTS + 1
    A = np.stack([np.stack([ring.fill_uniform(5, seed * 100 + i * 10 + b) for b in range(nbr)]) for i in range(s)])
    got, sm, sq = ctx.matmul_stream(A, 5, 5, geno, flags=flags, want_sums=True)
    want, wsm, wsq = ol.matmult4stream(ring, keys, 2.0 ** 34, A, 5, 5, geno, compute_sqsum=True, enc_prec=1)
    return got, want, (sm, sq), (wsm, wsq)


def test_one_by_one_matrix(env):
    """a single genotype: one live diagonal, baby step 0, giant step 0 - no rotation key is needed at all"""
    got, want, sums, wsums = product(env, np.array([[2]], dtype=np.int8), 1, 1)
    assert np.array_equal(got, want)
    assert np.array_equal(sums[0], wsums[0]) and np.array_equal(sums[1], wsums[1])


def test_all_missing_equals_all_zero(env):
    """missing genotypes (-1) contribute 0 to the product (matmult.go:1292-1295): an all-missing matrix gives the all-zero matrix's
    ciphertexts (every plaintext is the zero polynomial), and those match the oracle"""
    miss = np.full((7, 5), -1, dtype=np.int8)
    got_m, want_m, _, _ = product(env, miss, 2, 2)
    got_z, _, _, _ = product(env, np.zeros((7, 5), dtype=np.int8), 2, 2)
    assert np.array_equal(got_m, want_m)
    assert np.array_equal(got_m, got_z)


def test_single_row_and_single_column(env):
    """1 x 9 and 9 x 1 operands: the ragged extremes of GetDiag (one live entry per diagonal)"""
    rnd = np.random.default_rng(5)
    for shape, seed in (((1, 9), 3), ((9, 1), 4)):
        got, want, sums, wsums = product(env, rnd.integers(-1, 3, shape).astype(np.int8), 1, seed)
        assert np.array_equal(got, want), shape
        assert np.array_equal(sums[0], wsums[0]) and np.array_equal(sums[1], wsums[1])


def test_zero_length_batches_are_no_ops(env):
    """nct = 0 / nshift = 0 / n = 0: success, nothing written, no launch with an empty grid"""
    from sfgwas_amd import capi
    ctx, ring, _ = env
    L = capi.lib()
    guard = np.arange(64, dtype=np.uint64)
    d = ctx.to_device(guard); d2 = ctx.to_device(guard)
    assert L.sfg_ct_add_dev(ctx.h, d, d, d, 0, 5) == 0
    assert L.sfg_ct_sub_dev(ctx.h, d, d, d, 0, 5) == 0
    assert L.sfg_ct_rescale_dev(ctx.h, d, d, 0, 5) == 0
    assert L.sfg_ct_drop_level_dev(ctx.h, d, d, 0, 5, 3) == 0
    assert L.sfg_rotate_right_dev(ctx.h, d, d2, 0, 5, (C.c_int * 1)(0)) == 0
    assert L.sfg_encode_diags_dev(ctx.h, d, 8, 2, 2, 0, 0, 0, 5, d2) == 0
    one = np.ones(2, dtype=np.uint64)
    assert L.sfg_beaver_elem(ctx.h, 1, 2, capi.p64(np.array([(1 << 64) - 1, (1 << 63) - 1], dtype=np.uint64)), capi.p64(one), capi.p64(one), capi.p64(one), capi.p64(one), capi.p64(one), 0) == 0
    ctx.sync()
    assert np.array_equal(ctx.to_host(d, (64,), np.uint64), guard) and np.array_equal(ctx.to_host(d2, (64,), np.uint64), guard)
    ctx.free(d); ctx.free(d2)


def test_missing_rotation_key_is_an_error_not_a_wrong_answer(env):
    """RotateRightWithEvaluator log.Fatals on a missing key (basics.go:201-210); the C-ABI returns an error and names the rotation"""
    from sfgwas_amd import capi
    ctx, ring, _ = env
    ct = ring.fill_uniform(5, 77)[None]
    with pytest.raises(capi.SfgError) as e:
        ctx.rotate_right(ct, 5, [4099])                     # no test loads the key of right-rotation 4099
    assert "key" in str(e.value).lower()
    # the context stays usable
    out = ctx.rotate_right(ct, 5, [0])
    assert np.array_equal(out, ct)


def test_level_out_of_range_is_rejected(env):
    from sfgwas_amd import capi
    ctx, ring, _ = env
    L = capi.lib()
    nq = len(ol.Q_PN14)
    d = ctx.to_device(np.zeros(2 * (nq + 2) * ring.N, dtype=np.uint64))   # large enough for one ciphertext at any level a missing check could let through
    assert L.sfg_ct_add_dev(ctx.h, d, d, d, 1, nq) != 0                    # level == nq
    assert L.sfg_ct_add_dev(ctx.h, d, d, d, -1, 2) != 0
    assert L.sfg_fill_uniform_ct_dev(ctx.h, d, 1, nq, 1) != 0
    assert L.sfg_ct_rescale_dev(ctx.h, d, d, 1, 0) != 0                   # "cannot Rescale: already at level 0"
    assert L.sfg_encode_diags_dev(ctx.h, d, 8, 2, 2, 0, 0, 1, nq + 1, d) != 0
    ctx.free(d)
