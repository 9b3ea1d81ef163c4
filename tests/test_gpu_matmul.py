"""GPU parity of the whole product (sfg_matmul_stream = MatMult4Stream, matmult.go:1238-1505) against the
oracle's restatement: rotation cache, on-the-fly diagonal encode, lazy MAC, REDC, giant alignment, sum.
Every output word must match.  Rotation keys are uniform random words: parity of the arithmetic does not need
a valid key (decryption correctness of the same code path is covered at small N by tests/test_oracle_pinning.py)."""
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


def ensure_keys(ctx, ring, keys, rots_left):
    from sfgwas_amd import capi
    for k in rots_left:
        g = ring.galois(k)
        if g not in keys.keys:
            key = capi.random_rotkey(ring.moduli, ring.beta, ring.N, 1000 + k)
            keys.add(g, key)
            ctx.load_rotkey(g, key)


def active_steps(nrow, ncol):
    """baby / giant steps MatMult4Stream touches for an nrow x ncol operand (matmult.go:1329-1336)"""
    babies, giants = set(), set()
    nbr, m_ct = (nrow - 1) // SLOTS + 1, (ncol - 1) // SLOTS + 1
    for bi in range(nbr):
        nr = min((bi + 1) * SLOTS, nrow) - bi * SLOTS
        for bj in range(m_ct):
            nc = min((bj + 1) * SLOTS, ncol) - bj * SLOTS
            shifts = set(range(0, nr)) | set(range(SLOTS - nc + 1, SLOTS)) if nr + nc <= SLOTS else set(range(SLOTS))
            for sh in shifts:
                babies.add(sh % D)
                giants.add(sh // D)
    return sorted(babies), sorted(giants)


def run_case(env, nrow, ncol, s, in_level, max_level, flags, seed):
    from sfgwas_amd import capi
    ctx, ring, keys = env
    rnd = np.random.default_rng(seed)
    geno = rnd.integers(-1, 3, (nrow, ncol)).astype(np.int8)
    transposed = bool(flags & capi.SFG_TRANSPOSE)
    logical = np.ascontiguousarray(geno.T) if transposed else geno
    lrow, lcol = logical.shape
    babies, giants = active_steps(lrow, lcol)
    ensure_keys(ctx, ring, keys, [b for b in babies if b] + [g * D for g in giants if g])
    nbr = (lrow - 1) // SLOTS + 1
    A = np.stack([np.stack([ring.fill_uniform(in_level, seed * 100 + i * 10 + b) for b in range(nbr)]) for i in range(s)])
    got, sm, sq = ctx.matmul_stream(A, in_level, max_level, geno, flags=flags, want_sums=True)
    want, wsm, wsq = ol.matmult4stream(ring, keys, 2.0 ** 34, A, in_level, max_level, logical,
                                      compute_sqsum=True, square=bool(flags & capi.SFG_SQUARE), enc_prec=1)
    assert got.shape == want.shape
    assert np.array_equal(got, want), f"ciphertext words differ: {np.argwhere(got != want)[:5]}"
    if not transposed:
        assert np.array_equal(sm, wsm) and np.array_equal(sq, wsq)


def test_single_small_block(env):
    run_case(env, 60, 40, 2, 5, 5, 0, 1)


def test_two_block_columns_ragged_and_square(env):
    from sfgwas_amd import capi
    run_case(env, 30, SLOTS + 25, 1, 5, 5, capi.SFG_SQUARE, 2)


def test_transposed_operand_two_block_rows(env):
    from sfgwas_amd import capi
    run_case(env, 20, SLOTS + 10, 1, 5, 5, capi.SFG_TRANSPOSE, 3)


def test_level_drop(env):
    run_case(env, 25, 25, 1, 7, 5, 0, 4)


@pytest.mark.parametrize("s,in_level,max_level", [(1, 5, 3), (3, 4, 2), (17, 5, 5), (13, 5, 4)])
def test_fewer_moduli_and_row_counts(env, s, in_level, max_level):
    """max_level < 5 (three / two / four of the product's moduli: the int8 MAC's modulus indexing, with and without the 46-bit row beside it) and row counts
    that are neither 30 nor a multiple of 16 (s = 17: two row passes of 30 + 4; s = 13: the association scan's 26 rows; s = 3: one partly filled row tile)"""
    run_case(env, 70, 50, s, in_level, max_level, 0, 40 + s)


def _ctx_with_q0(q0):
    from sfgwas_amd import capi
    moduli = [q0] + list(ol.Q_PN14[1:])
    ctx = capi.Context(moduli, ol.P_PN14)
    ring = ol.Ring(14, moduli, ol.P_PN14)
    return ctx, ring, ol.RotKeys(ring)


@pytest.mark.parametrize("which", ["widest six-digit prime (int8 MAC, two-step Horner)", "first prime below 2^47 (fp64 MAC)"])
@pytest.mark.parametrize("nrow,ncol,s,flags", [(70, 50, 3, 0), (40, SLOTS + 30, 2, 2)])
def test_whole_product_with_a_47_bit_q0(which, nrow, ncol, s, flags):
    """every kernel of a product - key switch, encode FFT + NTT, MAC + epilogue, untile, giant alignment - on a modulus in
    [2^46, 2^47), every output word vs the oracle (matmult.go:1238-1505).  Up to 0x7F7F7F7F7F80 the 46/47-bit modulus multiplies on the int8 matrix core
    (six digit planes, k_mac_i8_ring<6, 2, 0, 2>, two-step Horner); above, a canonical word no longer fits six signed digits and the context keeps
    that modulus on the fp64 kernel k_mac_bc<true>."""
    from test_gpu_mac_i8 import widest_six_digit_prime, I8_BIG_QMAX
    q0 = widest_six_digit_prime() if which.startswith("widest") else ol.small_primes(14, 47, 1)[0]
    assert (q0 <= I8_BIG_QMAX) == which.startswith("widest") and (1 << 46) < q0 < (1 << 47)
    env_q = _ctx_with_q0(q0)
    try:
        run_case(env_q, nrow, ncol, s, 5, 5, flags, 470 + s)
    finally:
        env_q[0].close()
