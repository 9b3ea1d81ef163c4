"""GPU parity of the lazy MAC (sfg_mac_dev) against the reference's own arithmetic path restated in the
oracle: MForm(pt) -> MulCoeffsAndAdd128 -> ReduceAndAddUint128 -> eval.Reduce (matmult.go:247-440). Bit-exact."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
L_ = ol.lib


@pytest.fixture(scope="module")
def env():
    from sfgwas_amd import capi
    ctx = capi.Context(ol.Q_PN14, ol.P_PN14)
    yield ctx
    ctx.close()


def reference_mac(rot, pt, moduli, out_init=None):
    K, R, L, N = rot.shape
    Ncols = pt.shape[1]
    out = np.zeros((Ncols, R, L, N), dtype=np.uint64)
    ptm = pt.copy()
    for l in range(L):
        for k in range(K):
            for n in range(Ncols):
                L_().orc_mform_vec(ol.p64(ptm[k, n, l]), N, moduli[l])          # ToMontgomeryForm, :401-409
    for n in range(Ncols):
        for r in range(R):
            for l in range(L):
                acc = np.zeros((N, 2), dtype=np.uint64)
                for k in range(K):
                    L_().orc_mul_coeffs_and_add128(ol.p64(rot[k, r, l]), ol.p64(ptm[k, n, l]), ol.p64(acc), N)
                o = np.zeros(N, dtype=np.uint64)
                L_().orc_reduce_and_add_uint128(ol.p64(acc), ol.p64(o), L_().orc_mred_params(moduli[l]), moduli[l], N)
                L_().orc_canonical_reduce(ol.p64(o), N, moduli[l])
                if out_init is not None:
                    o = (o + out_init[n, r, l]) % np.uint64(moduli[l])
                out[n, r, l] = o
    return out


def rand_rows(rnd, shape_prefix, L, moduli, N):
    a = np.zeros(tuple(shape_prefix) + (L, N), dtype=np.uint64)
    for l in range(L):
        a[..., l, :] = rnd.integers(0, moduli[l], tuple(shape_prefix) + (N,), dtype=np.uint64)
    return a


@pytest.mark.parametrize("K,R,Ncols,L", [(20, 6, 5, 5), (9, 30, 33, 5), (3, 32, 2, 2), (1, 1, 1, 1), (91, 4, 3, 6), (37, 26, 7, 5), (13, 10, 3, 5), (6, 22, 35, 5), (5, 17, 2, 6)])
def test_mac_random_bit_exact(env, K, R, Ncols, L):
    ctx = env
    rnd = np.random.default_rng(K * 1000 + R)
    N = ctx.N
    rot = rand_rows(rnd, (K, R), L, ol.Q_PN14, N)
    pt = rand_rows(rnd, (K, Ncols), L, ol.Q_PN14, N)
    got = ctx.mac(rot, pt, L)
    want = reference_mac(rot, pt, ol.Q_PN14)
    assert np.array_equal(got, want)


def test_mac_worst_case_magnitudes_and_accumulate(env):
    """all operands q-1 over K=150 terms exercises the exact-run bound and the periodic fold; accumulate adds onto out."""
    ctx = env
    K, R, Ncols, L, N = 150, 3, 2, 5, ctx.N
    rot = np.zeros((K, R, L, N), dtype=np.uint64)
    pt = np.zeros((K, Ncols, L, N), dtype=np.uint64)
    for l in range(L):
        rot[:, :, l, :] = ol.Q_PN14[l] - 1
        pt[:, :, l, :] = ol.Q_PN14[l] - 1
    rnd = np.random.default_rng(3)
    init = rand_rows(rnd, (Ncols, R), L, ol.Q_PN14, N)
    got = ctx.mac(rot, pt, L, out_init=init)
    for l in range(L):
        q = ol.Q_PN14[l]
        val = (K * (q - 1) * (q - 1)) % q
        want = (init[:, :, l, :].astype(object) + val) % q
        assert np.array_equal(got[:, :, l, :].astype(object), want)


def test_mac_centred_operand_and_packed_limb_worst_case(env):
    """The device works with rot centred to (-q/2, q/2] and plaintext limbs biased by 4096: the largest single term is rot = (q-1)/2 (or its negative,
    (q+1)/2) against all-ones 12-bit limbs, every term of one sign.  K = 200 spans several flush periods (60 k-steps for 35-bit moduli)."""
    ctx = env
    K, R, Ncols, L, N = 200, 5, 3, 5, ctx.N
    rot = np.zeros((K, R, L, N), dtype=np.uint64)
    pt = np.zeros((K, Ncols, L, N), dtype=np.uint64)
    for l in range(L):
        q = ol.Q_PN14[l]
        hi = ((q >> 24) - 1) << 24 | 0xFFFFFF            # largest residue with all-ones low limbs
        assert hi < q
        rot[:, 0::2, l, :] = (q - 1) // 2                # +max after centring
        rot[:, 1::2, l, :] = (q + 1) // 2                # -max after centring
        pt[:, :, l, :] = hi
        pt[:, 1, l, : N // 2] = q - 1
    got = ctx.mac(rot, pt, L)
    for l in range(L):
        q = ol.Q_PN14[l]
        for r in range(R):
            for n in range(Ncols):
                for half, sl in ((0, slice(0, N // 2)), (1, slice(N // 2, N))):
                    want = K * int(rot[0, r, l, 0]) * int(pt[0, n, l, 0 if half == 0 else N - 1]) % q
                    assert (got[n, r, l, sl] == want).all(), (l, r, n, half)


def test_mac_with_a_47_bit_modulus():
    """The Karatsuba middle term (r_lo + r_hi)(p_lo + p_hi) reaches 2.25 * 2^48 for q in [2^46, 2^47): the fold period must follow the actual modulus.
    q0 just below 2^47, all operands q - 1 (and random ones), K spanning many fold periods, both MAC kernels' paths through sfg_mac_dev."""
    from sfgwas_amd import capi
    q47 = ol.small_primes(14, 47, 1)[0]
    assert (1 << 46) <= q47 < (1 << 47)
    moduli = [q47] + list(ol.Q_PN14[1:3])
    ctx = capi.Context(moduli, ol.P_PN14)
    K, R, Ncols, L, N = 240, 3, 2, 2, ctx.N
    rnd = np.random.default_rng(47)
    rot = rand_rows(rnd, (K, R), L, moduli, N)
    pt = rand_rows(rnd, (K, Ncols), L, moduli, N)
    rot[:, 0, 0, :] = q47 - 1; pt[:, 0, 0, :] = q47 - 1                   # worst case on (row 0, column 0) of the big modulus
    rot[:, 1, 0, : N // 2] = (1 << 46) + (1 << 23) - 1                       # hi and lo halves both maximal
    got = ctx.mac(rot, pt, L)
    for l in range(L):
        q = moduli[l]
        for r in range(R):
            for n in range(Ncols):
                want = np.array([sum(int(a) * int(b) for a, b in zip(rot[:, r, l, x], pt[:, n, l, x])) % q for x in (0, 1, N // 2, N - 1)], dtype=object)
                assert [int(v) for v in got[n, r, l, [0, 1, N // 2, N - 1]]] == list(want), (l, r, n)
    # whole rows of the worst-case pair
    want00 = (K * (q47 - 1) * (q47 - 1)) % q47
    assert (got[0, 0, 0, :] == np.uint64(want00)).all()
    ctx.close()
