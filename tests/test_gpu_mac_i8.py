"""GPU parity of the DEFAULT ring MAC - the int8 matrix-core kernels of sfgwas_amd/csrc/mac_i8.hip (k_i8_pack_rot, k_i8_pack_pt_digits / k_i8_pack_pt,
k_mac_i8_ring<5, 3, 0, 2> / <6, 2, 0, 2>, k_mac_i8<ND>, k_i8_untile) - reached through the test hook sfg_mac_i8_dev exactly as a product reaches them, against the
reference's own arithmetic chain restated in the oracle: MForm(pt) -> MulCoeffsAndAdd128 -> ReduceAndAddUint128 -> eval.Reduce (gwas/matmult.go:247-440), and
against Python integers.  Bit-exact, every coefficient of the checked (column, row) pairs.

The kernels serve mirror-symmetric plaintexts (P[N-1-x] = P[x]: what a real slot vector encodes to) from half rows; the test builds such rows.
Long contractions repeat P distinct k-slices cyclically on the device (capi.Context.mac_i8), so the expected sum is sum_p cnt_p rot_p pt_p - the multiplicity is
folded into the oracle's plaintext operand (cnt_p pt_p mod q), the chain itself is the reference's."""
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


@pytest.fixture(scope="module")
def env47():
    """q0 a 47-bit prime, the widest one whose words fit six signed digits: the two-step Horner recombination"""
    from sfgwas_amd import capi
    q47 = widest_six_digit_prime()
    moduli = [q47] + list(ol.Q_PN14[1:3])
    ctx = capi.Context(moduli, ol.P_PN14)
    yield ctx, moduli
    ctx.close()


I8_BIG_QMAX = 0x7F7F7F7F7F80          # common.hpp SFG_I8_BIG_QMAX: canonical words up to q - 1 must fit six signed digits (<= 0x7F7F7F7F7F7F)


def widest_six_digit_prime():
    """the largest NTT-friendly prime (== 1 mod 2N) the int8 MAC takes: in (2^46 - 2^24, I8_BIG_QMAX], so the Horner recombination runs its two-step form"""
    from sympy import isprime
    M = 2 << 14
    x = I8_BIG_QMAX - (I8_BIG_QMAX - 1) % M
    while not isprime(x):
        x -= M
    assert (1 << 46) < x <= I8_BIG_QMAX and x % M == 1
    return x


def sdigits(v, nd):
    """signed base-256 digits as i8_digits<ND> (mac_i8.hip) takes them"""
    out = []
    for _ in range(nd):
        lo = ((v + 128) & 255) - 128
        out.append(lo)
        v = (v - lo) >> 8
    assert v == 0
    return out


def extreme_words(q, nd):
    """canonical words whose digits sit at the ends of the int8 range: (all lower digits -128, all lower digits +127), top digit as large as q allows"""
    lo_neg = sum(128 << (8 * i) for i in range(nd - 1))
    lo_pos = sum(127 << (8 * i) for i in range(nd - 1))
    t = (q - 1 + lo_neg) >> (8 * (nd - 1))
    while (t << (8 * (nd - 1))) - lo_neg >= q:
        t -= 1
    a = (t << (8 * (nd - 1))) - lo_neg
    t2 = (q - 1 - lo_pos) >> (8 * (nd - 1))
    b = (t2 << (8 * (nd - 1))) + lo_pos
    assert 0 <= a < q and 0 <= b < q
    assert sdigits(a, nd)[:nd - 1] == [-128] * (nd - 1) and sdigits(b, nd)[:nd - 1] == [127] * (nd - 1)
    return a, b


def build_operands(rnd, P, R, Ncols, L, moduli, N):
    """P random k-slices with the worst cases planted on fixed rows / columns of EVERY slice (so that they meet in every k-step):
       rot row 0: q-1;  row 1: (q-1)/2 (largest centred value);  row 2: (q+1)/2 (its negative);  row 3: centred value with all lower digits -128;  row 4: ... +127;  row 5: its negative
       pt  col 0: q-1;  col 1: canonical word with all lower digits -128;  col 2: ... +127;  col 3: 0"""
    H = N // 2
    rot = np.zeros((P, R, L, N), dtype=np.uint64)
    pt = np.zeros((P, Ncols, L, H), dtype=np.uint64)
    for l in range(L):
        q = int(moduli[l])
        nd = 6 if q >= (1 << 36) else 5
        rot[:, :, l, :] = rnd.integers(0, q, (P, R, N), dtype=np.uint64)
        pt[:, :, l, :] = rnd.integers(0, q, (P, Ncols, H), dtype=np.uint64)
        a, b = extreme_words(q // 2, nd)                 # magnitudes <= q/2 for the centred operand
        ea, eb = extreme_words(q, nd)
        plant_r = [q - 1, (q - 1) // 2, (q + 1) // 2, a, b, q - b]     # (a, b < q/2 are their own centred values; q - b centres to -b)
        plant_c = [q - 1, ea, eb, 0]
        for i, v in enumerate(plant_r[:R]):
            rot[:, i, l, :] = v
        for j, v in enumerate(plant_c[:Ncols]):
            pt[:, j, l, :] = v
    return rot, pt


def reference(rot, pt_half, moduli, K, cols, out_init=None):
    """the reference chain on the checked columns: out[n][r][l][:] for n in cols"""
    P, R, L, N = rot.shape
    cnt = [K // P + (1 if p < K % P else 0) for p in range(P)]
    out = {}
    for n in cols:
        full = np.concatenate([pt_half[:, n], pt_half[:, n, :, ::-1]], axis=-1)           # [P][L][N]: P[N-1-x] = P[x]
        ptm = np.zeros_like(full)
        for l in range(L):
            q = int(moduli[l])
            for p in range(P):
                ptm[p, l] = (full[p, l].astype(object) * cnt[p] % q).astype(np.uint64)       # multiplicity of slice p folded into the plaintext
                L_().orc_mform_vec(ol.p64(ptm[p, l]), N, q)                               # ToMontgomeryForm, matmult.go:401-409
        res = np.zeros((R, L, N), dtype=np.uint64)
        for r in range(R):
            for l in range(L):
                q = int(moduli[l])
                acc = np.zeros((N, 2), dtype=np.uint64)
                for p in range(P):
                    L_().orc_mul_coeffs_and_add128(ol.p64(np.ascontiguousarray(rot[p, r, l])), ol.p64(ptm[p, l]), ol.p64(acc), N)     # :247-289
                o = np.zeros(N, dtype=np.uint64)
                L_().orc_reduce_and_add_uint128(ol.p64(acc), ol.p64(o), L_().orc_mred_params(q), q, N)                                  # :291-324
                L_().orc_canonical_reduce(ol.p64(o), N, q)
                if out_init is not None:
                    o = (o + out_init[n, r, l]) % np.uint64(q)
                res[r, l] = o
        out[n] = res
    return out


def check_planted(got, rot, pt_half, moduli, K, rows, cols):
    """the planted worst cases once more against plain Python integers (constant operands: K a b mod q)"""
    N = rot.shape[-1]
    for l, q in enumerate(moduli[:rot.shape[2]]):
        q = int(q)
        for r in rows:
            for n in cols:
                a, b = int(rot[0, r, l, 0]), int(pt_half[0, n, l, 0])
                assert (rot[:, r, l, :] == a).all() and (pt_half[:, n, l, :] == b).all()
                assert (got[n, r, l, :] == np.uint64(K * a * b % q)).all(), (l, r, n)


CASES = [
    # K, P, R, Ncols, L, form  - Ncols 91 (six column tiles): the LDS-ring kernels; 33: the cache-shared kernels; form 1: panel words (k_i8_pack_pt)
    (64, 5, 30, 91, 5, 0),
    (1183, 7, 30, 91, 2, 0),
    (1456, 5, 26, 91, 2, 0),
    (2184, 5, 30, 91, 2, 0),
    (200, 7, 30, 33, 3, 0),
    (91, 7, 10, 91, 2, 1),
    (150, 5, 32, 33, 2, 1),
]


@pytest.mark.parametrize("K,P,R,Ncols,L,form", CASES)
def test_default_mac_matches_the_reference_chain(env, K, P, R, Ncols, L, form):
    ctx = env
    N = ctx.N
    rnd = np.random.default_rng(K * 131 + Ncols)
    rot, pt = build_operands(rnd, P, R, Ncols, L, ol.Q_PN14, N)
    got = ctx.mac_i8(rot, pt, L, K=K, pt_form=form)
    cols = sorted({0, 1, 2, 3, 15, 16, Ncols // 2, Ncols - 1})
    want = reference(rot, pt, ol.Q_PN14, K, cols)
    for n in cols:
        assert np.array_equal(got[n], want[n]), (n,)
    check_planted(got, rot, pt, ol.Q_PN14, K, range(min(R, 6)), range(4))


def test_default_mac_accumulates_onto_out(env):
    ctx = env
    K, P, R, Ncols, L, N = 300, 5, 30, 91, 2, ctx.N
    rnd = np.random.default_rng(9)
    rot, pt = build_operands(rnd, P, R, Ncols, L, ol.Q_PN14, N)
    init = np.zeros((Ncols, R, L, N), dtype=np.uint64)
    for l in range(L):
        init[:, :, l, :] = rnd.integers(0, ol.Q_PN14[l], (Ncols, R, N), dtype=np.uint64)
    init[0, 0, :, :] = np.array([q - 1 for q in ol.Q_PN14[:L]], dtype=np.uint64)[:, None]
    got = ctx.mac_i8(rot, pt, L, K=K, out_init=init)
    cols = [0, 5, 90]
    want = reference(rot, pt, ol.Q_PN14, K, cols, out_init=init)
    for n in cols:
        assert np.array_equal(got[n], want[n]), (n,)


@pytest.mark.parametrize("K,P,R,Ncols,form", [(1456, 5, 30, 91, 0), (300, 7, 30, 33, 0), (182, 5, 12, 91, 1)])
def test_default_mac_with_a_47_bit_modulus(env47, K, P, R, Ncols, form):
    """q0 in (2^46, 0x7F7F7F7F7F80]: one Horner step r 256 + D passes 2^53 there (128 q), the kernels take it as two x 16 steps (i8_horner, mac_i8.hip).
    Round 4's epilogue gave 18 468 wrong words of 20 000 on such a prime."""
    ctx, moduli = env47
    L, N = 2, ctx.N
    rnd = np.random.default_rng(K + 47)
    rot, pt = build_operands(rnd, P, R, Ncols, L, moduli, N)
    got = ctx.mac_i8(rot, pt, L, K=K, pt_form=form)
    cols = sorted({0, 1, 2, 3, 16, Ncols - 1})
    want = reference(rot, pt, moduli, K, cols)
    for n in cols:
        assert np.array_equal(got[n], want[n]), (n,)
    check_planted(got, rot, pt, moduli, K, range(min(R, 6)), range(4))


def test_hook_is_refused_without_the_test_switch(monkeypatch):
    from sfgwas_amd import capi
    monkeypatch.setenv("SFG_ENABLE_TEST_HOOKS", "0")
    ctx = capi.Context(ol.Q_PN14, ol.P_PN14)
    try:
        with pytest.raises(capi.SfgError, match="test hook"):
            ctx.mac_i8(np.zeros((1, 1, 1, ctx.N), dtype=np.uint64), np.zeros((1, 1, 1, ctx.N // 2), dtype=np.uint64), 1)
    finally:
        ctx.close()
