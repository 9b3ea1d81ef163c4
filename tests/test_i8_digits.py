"""The arithmetic of the int8 matrix-core MAC (sfgwas_amd/csrc/mac_i8.hip), restated with Python integers - no GPU:
  * balanced base-256 digits through the bias trick (the bytes of v + 0x80..80 with their top bits flipped) reproduce v, for canonical plaintext words
    (< 2^36, five digits; < 2^46, six) and centred rot words;
  * the ND^2 digit products of K k-steps, summed per weight a + b in int32, never overflow for the K the launch accepts (worst-case digits);
  * Horner mod q over the 2 ND - 1 sums with the fp64 reduction the kernel uses equals the ring MAC sum(pt * rot) mod q of matmult.go:247-289."""
import numpy as np
import pytest

Q35, Q46 = 34359214081, 35184376545281


def digits(v, nd):
    bias = sum(0x80 << (8 * i) for i in range(nd))
    t = v + bias
    assert 0 <= t < 1 << (8 * nd)
    return [((t >> (8 * i)) & 0xFF) ^ 0x80 for i in range(nd)]          # two's-complement bytes of the signed digits


def signed(b):
    return b - 256 if b >= 128 else b


@pytest.mark.parametrize("q,nd", [(Q35, 5), (Q46, 6)])
def test_bias_trick_gives_balanced_digits(q, nd):
    rnd = np.random.default_rng(nd)
    vals = [0, 1, q - 1, q // 2, -(q // 2), (q // 2) + 1 - q] + [int(x) for x in rnd.integers(0, q, 200)] + [int(x) - q // 2 for x in rnd.integers(0, q, 200)]
    for v in vals:
        d = [signed(b) for b in digits(v, nd)]
        assert all(-128 <= x <= 127 for x in d)
        assert sum(x << (8 * i) for i, x in enumerate(d)) == v


@pytest.mark.parametrize("q,nd,K", [(Q35, 5, 2184), (Q46, 6, 2184)])
def test_digit_sums_fit_int32_and_horner_recovers_the_ring_mac(q, nd, K):
    rnd = np.random.default_rng(K + nd)
    assert K * nd < 131072                                               # the launch's guard: nd products of at most 2^14 per k-step and weight
    for trial in range(3):
        if trial == 0:                                                   # worst case for the sums: every digit -128
            pt = np.full(K, -sum(128 << (8 * i) for i in range(nd)), dtype=object); rot = pt.copy()
        else:
            pt = np.array([int(x) for x in rnd.integers(0, q, K)], dtype=object)                   # canonical plaintext words
            rot = np.array([int(x) - q // 2 for x in rnd.integers(0, q, K)], dtype=object)         # centred rot words
        D = [0] * (2 * nd - 1)
        for k in range(K):
            a = [signed(b) for b in digits(int(rot[k]), nd)] if trial else [-128] * nd
            b = [signed(x) for x in digits(int(pt[k]), nd)] if trial else [-128] * nd
            for i in range(nd):
                for j in range(nd):
                    D[i + j] += a[i] * b[j]
        assert all(abs(x) < 2 ** 31 for x in D)
        # the kernel's epilogue in fp64: r = D[top]; r = r * 256 + D[s] reduced by q * rint(x / q) - every intermediate is an exact integer below 2^53
        r = float(D[-1])
        qf, qinv = float(q), 1.0 / float(q)
        for s in range(2 * nd - 3, -1, -1):
            x = r * 256.0 + float(D[s])
            assert abs(x) < 2 ** 53
            r = x - qf * np.rint(x * qinv)
        if r < 0:
            r += qf
        want = sum(int(p) * int(t) for p, t in zip(pt, rot)) % q
        assert int(r) % q == want


def test_horner_needs_two_steps_of_16_above_2_46():
    """The one-step recombination x = r 256 + D is exact only while 128 q + 2^31 < 2^53 (q <= 2^46 - 2^24).  For a 47-bit prime - accepted by sfg_ctx_create - the
    kernels split the step (i8_horner, mac_i8.hip): replayed here in numpy float64 (no FMA, as the library is built with -ffp-contract=off) on random digit sums:
    the two-step form equals integer arithmetic everywhere, the one-step form does not (that was round 4's latent wrong-answer path)."""
    q = (1 << 47) - 229375                       # 0x7ffffffc8001, prime, == 1 mod 2^15
    assert q % (1 << 15) == 1 and q > (1 << 46)
    nd, n = 6, 20000
    rnd = np.random.default_rng(46)
    D = rnd.integers(-(2 ** 31) + 1, 2 ** 31, (2 * nd - 1, n))
    want = np.array([sum(int(D[s, i]) << (8 * s) for s in range(2 * nd - 1)) % q for i in range(n)], dtype=object)
    qf, qinv = np.float64(q), np.float64(1.0) / np.float64(q)

    def horner(two_step):
        r = D[-1].astype(np.float64)
        for s in range(2 * nd - 3, -1, -1):
            if two_step:
                x1 = r * 16.0
                assert (np.abs(x1) < 2.0 ** 53).all()
                r1 = x1 - qf * np.rint(x1 * qinv)
                x = r1 * 16.0 + D[s].astype(np.float64)
            else:
                x = r * 256.0 + D[s].astype(np.float64)
            r = x - qf * np.rint(x * qinv)
        r = np.where(r < 0, r + qf, r)
        return np.array([int(v) for v in r], dtype=object)

    assert (horner(True) == want).all()
    assert (horner(False) != want).sum() > n // 2          # the defect the split removes
    # and the PN14QP438 prime stays on the one-step form, exactly
    q0 = Q46
    want0 = np.array([sum(int(D[s, i]) << (8 * s) for s in range(2 * nd - 1)) % q0 for i in range(n)], dtype=object)
    qf, qinv = np.float64(q0), np.float64(1.0) / np.float64(q0)
    assert q0 <= (1 << 46) - (1 << 24)
    assert (horner(False) == want0).all()
