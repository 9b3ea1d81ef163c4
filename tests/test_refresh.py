"""Collective bootstrap, local work (mpc/mhe.go:222-348 -> dckks.RefreshProtocol GenShares / Decrypt / Recode / Recrypt).

PARITY UNPINNED: the protocol is restated from the published lattigo v2.1.0 dckks/refresh.go (the fork's source is absent).  The CPU tests pin
the oracle against Python big integers (big-int -> RNS, CRT + recentring) and against the protocol's own contract: two parties refresh a
ciphertext encrypted under the sum of their shards and the result decrypts, at the top level, to the same message plus the known error
terms.  The GPU tests compare every word of the HIP path with the oracle."""
import numpy as np
import pytest

import oracle_lib as ol


def centred(v, q):
    v = int(v) % q
    return v - q if v > q // 2 else v


def crt(residues, moduli):
    Q = 1
    for q in moduli:
        Q *= q
    x = 0
    for r, q in zip(residues, moduli):
        Qi = Q // q
        x += int(r) * pow(Qi, -1, q) * Qi
    return x % Q, Q


def make_masks(rnd, n, bound, W):
    vals = []
    for _ in range(n):
        m = int.from_bytes(rnd.bytes(64), "little") % bound
        if m >= bound >> 1:
            m -= bound
        vals.append(m)
    return vals, ol.bigints_to_limbs(vals, W)


@pytest.fixture(scope="module")
def small_ring():
    q = ol.small_primes(10, 36, 4)
    p = ol.small_primes(10, 40, 1)
    return ol.Ring(10, q, p)


def test_bigint_to_rns_matches_python_ints(small_ring):
    ring = small_ring
    rnd = np.random.default_rng(5)
    W = 3
    vals = [0, 1, -1, (1 << 150) - 3, -(1 << 150) + 7] + [int.from_bytes(rnd.bytes(20), "little") - (1 << 159) for _ in range(ring.N - 5)]
    limbs = ol.bigints_to_limbs(vals, W)
    out = np.zeros((ring.nq, ring.N), dtype=np.uint64)
    ol.lib().orc_bigint_to_rns(ring.h, ring.nq, ol.p64(limbs), W, ol.p64(out))
    for j in range(ring.nq):
        q = ring.moduli[j]
        assert [int(x) for x in out[j]] == [v % q for v in vals]


def test_refresh_finish_is_crt_recentre_reduce(small_ring):
    """Recode against Python integers: x = CRT(INTT(c0 + h0)); x >= Q//2 -> x - Q; residues modulo every q_j; then NTT + h1, c1 = crs"""
    ring = small_ring
    level = 1
    ct = ring.fill_uniform(level, 41)
    rnd = np.random.default_rng(8)
    h0 = np.stack([rnd.integers(0, ring.moduli[j], ring.N, dtype=np.uint64) for j in range(level + 1)])
    h1 = np.stack([rnd.integers(0, ring.moduli[j], ring.N, dtype=np.uint64) for j in range(ring.nq)])
    crs = np.stack([rnd.integers(0, ring.moduli[j], ring.N, dtype=np.uint64) for j in range(ring.nq)])
    out = ol.refresh_finish(ring, level, ct, h0, h1, crs)
    x = [ring.intt(j, (ct[0, j] + h0[j]) % np.uint64(ring.moduli[j])) for j in range(level + 1)]
    mods = ring.moduli[:level + 1]
    want = np.zeros((ring.nq, ring.N), dtype=np.uint64)
    for c in range(ring.N):
        v, Q = crt([x[j][c] for j in range(level + 1)], mods)
        if v >= Q >> 1:
            v -= Q
        for j in range(ring.nq):
            want[j, c] = v % ring.moduli[j]
    for j in range(ring.nq):
        assert np.array_equal(out[0, j], (ring.ntt(j, want[j]) + h1[j]) % np.uint64(ring.moduli[j]))
        assert np.array_equal(out[1, j], crs[j])


def test_two_party_refresh_decrypts_to_the_same_message_at_the_top_level(small_ring):
    ring = small_ring
    level, W, nparties = 1, 2, 2
    s1, s2 = ring.gen_secret(1), ring.gen_secret(2)
    rnd = np.random.default_rng(3)
    m = rnd.integers(-(1 << 30), 1 << 30, ring.N)
    # a ciphertext under s = s1 + s2:  c1 uniform, c0 = m - s c1   (error-free, so the identity below is exact)
    sk1, sk2 = ol.secret_ntt(ring, s1), ol.secret_ntt(ring, s2)
    ct = ring.fill_uniform(level, 77)
    for j in range(level + 1):
        q = ring.moduli[j]
        mj = ring.ntt(j, np.array([int(v) % q for v in m], dtype=np.uint64))
        sc1 = np.array([(int(a) + int(b)) * int(c) % q for a, b, c in zip(sk1[j], sk2[j], ct[1, j])], dtype=np.uint64)
        ct[0, j] = (mj + np.uint64(q) - sc1) % np.uint64(q)
    Ql = 1
    for q in ring.moduli[:level + 1]:
        Ql *= q
    bound = Ql // (2 * nparties)
    crs = np.stack([rnd.integers(0, ring.moduli[j], ring.N, dtype=np.uint64) for j in range(ring.nq)])
    shares, errs = [], []
    for sk, seed in ((sk1, 11), (sk2, 12)):
        r2 = np.random.default_rng(seed)
        _, limbs = make_masks(r2, ring.N, bound, W)
        e0, e1 = r2.integers(-19, 20, ring.N).astype(np.int32), r2.integers(-19, 20, ring.N).astype(np.int32)
        shares.append(ol.refresh_gen_shares(ring, level, ct, sk, crs, limbs, e0, e1))
        errs.append((e0, e1))
    h0 = np.stack([(shares[0][0][j] + shares[1][0][j]) % np.uint64(ring.moduli[j]) for j in range(level + 1)])
    h1 = np.stack([(shares[0][1][j] + shares[1][1][j]) % np.uint64(ring.moduli[j]) for j in range(ring.nq)])
    out = ol.refresh_finish(ring, level, ct, h0, h1, crs)
    want = m + errs[0][0] + errs[1][0] - errs[0][1] - errs[1][1]
    for j in range(ring.nq):
        q = ring.moduli[j]
        dec = np.array([(int(a) + (int(b) + int(c)) * int(d)) % q for a, b, c, d in zip(out[0, j], sk1[j], sk2[j], out[1, j])], dtype=np.uint64)
        got = [centred(v, q) for v in ring.intt(j, dec)]
        assert got == [int(v) for v in want], f"modulus {j}"


# ---------------------------------------------------------------- the target-scale form the reference calls (mhe.go:315,330)
def quo(a, b):
    """big.Int.Quo: truncated towards zero"""
    return abs(a) // b * (1 if a >= 0 else -1)


SCALES = [(2.0 ** 68, 2.0 ** 34),                                   # a fresh product: A.scale * Delta -> Delta
          (2.0 ** 68 / 34359410689.0 * 2.0 ** 34, 2.0 ** 34),        # a product of a rescaled operand: not a power of two
          (2.0 ** 34, 2.0 ** 34),                                    # ratio 1: must reproduce the unscaled form
          (1234567.0 * 2.0 ** 20, 2.0 ** 40 + 2.0 ** 7)]             # arbitrary float64 scales, target above the input


@pytest.mark.parametrize("ct_scale,target", SCALES)
def test_scaled_shares_and_recode_match_python_big_integers(small_ring, ct_scale, target):
    """GenShares: h1 comes from Quo(mask * Int(target), Int(scale)) and h0 from the mask itself; Recode: Quo(x * Int(target), Int(scale)) on the centred
    CRT value (lattigo v2.2.0 dckks/refresh.go, restated; parity unpinned) - the oracle against Python ints, every coefficient"""
    ring = small_ring
    level, W = 2, 3
    oi, ii = int(target), int(ct_scale)
    rnd = np.random.default_rng(21)
    Ql = 1
    for q in ring.moduli[:level + 1]:
        Ql *= q
    vals, limbs = make_masks(rnd, ring.N, Ql // 4, W)
    edge = [0, 1, -1, (Ql // 4 >> 1) - 1, -(Ql // 4 >> 1), ii, -ii, ii - 1, -(ii - 1), ii + 1]
    vals[:len(edge)] = edge
    limbs[:len(edge)] = ol.bigints_to_limbs(edge, W)
    ct = ring.fill_uniform(level, 5)
    zero_sk = np.zeros((ring.nq, ring.N), dtype=np.uint64)
    zero_e = np.zeros(ring.N, dtype=np.int32)
    crs = np.stack([rnd.integers(0, ring.moduli[j], ring.N, dtype=np.uint64) for j in range(ring.nq)])
    h0, h1 = ol.refresh_gen_shares_scaled(ring, level, ct, ct_scale, target, zero_sk, crs, limbs, zero_e, zero_e)
    for j in range(level + 1):                                       # sk = 0, e = 0: h0 = NTT(mask)
        assert np.array_equal(h0[j], ring.ntt(j, np.array([v % ring.moduli[j] for v in vals], dtype=np.uint64)))
    scaled = [quo(v * oi, ii) for v in vals]
    for j in range(ring.nq):                                         # h1 = -NTT(scaled mask)
        q = ring.moduli[j]
        assert np.array_equal(h1[j], ring.ntt(j, np.array([(-v) % q for v in scaled], dtype=np.uint64)))
    if ct_scale == target:
        u0, u1 = ol.refresh_gen_shares(ring, level, ct, zero_sk, crs, limbs, zero_e, zero_e)
        assert np.array_equal(u0, h0) and np.array_equal(u1, h1)
    # Recode
    h0a = np.stack([rnd.integers(0, ring.moduli[j], ring.N, dtype=np.uint64) for j in range(level + 1)])
    h1a = np.stack([rnd.integers(0, ring.moduli[j], ring.N, dtype=np.uint64) for j in range(ring.nq)])
    out = ol.refresh_finish_scaled(ring, level, ct, ct_scale, target, h0a, h1a, crs)
    x = [ring.intt(j, (ct[0, j] + h0a[j]) % np.uint64(ring.moduli[j])) for j in range(level + 1)]
    want = np.zeros((ring.nq, ring.N), dtype=np.uint64)
    for c in range(ring.N):
        v, Q = crt([x[j][c] for j in range(level + 1)], ring.moduli[:level + 1])
        if v >= Q >> 1:
            v -= Q
        v = quo(v * oi, ii)
        for j in range(ring.nq):
            want[j, c] = v % ring.moduli[j]
    for j in range(ring.nq):
        assert np.array_equal(out[0, j], (ring.ntt(j, want[j]) + h1a[j]) % np.uint64(ring.moduli[j]))
        assert np.array_equal(out[1, j], crs[j])
    if ct_scale == target:
        assert np.array_equal(out, ol.refresh_finish(ring, level, ct, h0a, h1a, crs))


def test_two_party_refresh_from_scale_2_68_to_2_34_decrypts_to_the_rescaled_message(small_ring):
    """the protocol identity at the reference's scales: a level-1 ciphertext of m at scale 2^68 under s1 + s2, refreshed to scale 2^34, decrypts at the
    top level to m / 2^34 up to the truncation units of the two masks and the message (|error| <= 3) plus the known noise terms"""
    ring = small_ring
    level, W, nparties = 1, 2, 2
    ct_scale, target = 2.0 ** 68, 2.0 ** 34
    s1, s2 = ring.gen_secret(1), ring.gen_secret(2)
    rnd = np.random.default_rng(3)
    m = [int(v) << 34 for v in rnd.integers(-(1 << 20), 1 << 20, ring.N)]      # message at scale 2^68 (values of ~2^20 * 2^34 at Delta^2 ... here < 2^55)
    m = [v + int(rnd.integers(-(1 << 33), 1 << 33)) for v in m]
    sk1, sk2 = ol.secret_ntt(ring, s1), ol.secret_ntt(ring, s2)
    ct = ring.fill_uniform(level, 77)
    for j in range(level + 1):
        q = ring.moduli[j]
        mj = ring.ntt(j, np.array([v % q for v in m], dtype=np.uint64))
        sc1 = np.array([(int(a) + int(b)) * int(c) % q for a, b, c in zip(sk1[j], sk2[j], ct[1, j])], dtype=np.uint64)
        ct[0, j] = (mj + np.uint64(q) - sc1) % np.uint64(q)
    Ql = 1
    for q in ring.moduli[:level + 1]:
        Ql *= q
    bound = Ql // (2 * nparties)
    crs = np.stack([rnd.integers(0, ring.moduli[j], ring.N, dtype=np.uint64) for j in range(ring.nq)])
    shares, errs = [], []
    for sk, seed in ((sk1, 11), (sk2, 12)):
        r2 = np.random.default_rng(seed)
        _, limbs = make_masks(r2, ring.N, bound, W)
        e0, e1 = r2.integers(-19, 20, ring.N).astype(np.int32), r2.integers(-19, 20, ring.N).astype(np.int32)
        shares.append(ol.refresh_gen_shares_scaled(ring, level, ct, ct_scale, target, sk, crs, limbs, e0, e1))
        errs.append((e0, e1))
    h0 = np.stack([(shares[0][0][j] + shares[1][0][j]) % np.uint64(ring.moduli[j]) for j in range(level + 1)])
    h1 = np.stack([(shares[0][1][j] + shares[1][1][j]) % np.uint64(ring.moduli[j]) for j in range(ring.nq)])
    out = ol.refresh_finish_scaled(ring, level, ct, ct_scale, target, h0, h1, crs)
    for j in range(ring.nq):
        q = ring.moduli[j]
        dec = np.array([(int(a) + (int(b) + int(c)) * int(d)) % q for a, b, c, d in zip(out[0, j], sk1[j], sk2[j], out[1, j])], dtype=np.uint64)
        got = [centred(v, q) for v in ring.intt(j, dec)]
        for c in range(ring.N):
            ideal = (m[c] + int(errs[0][0][c]) + int(errs[1][0][c])) / 2.0 ** 34 - int(errs[0][1][c]) - int(errs[1][1][c])
            assert abs(got[c] - ideal) <= 3.0, (j, c, got[c], ideal)


@pytest.mark.gpu
@pytest.mark.parametrize("level,W,nct,ct_scale,target", [(4, 4, 2, 2.0 ** 68, 2.0 ** 34), (4, 4, 1, 2.0 ** 68 / 34359410689.0 * 2.0 ** 34, 2.0 ** 34),
                                                           (9, 6, 1, 2.0 ** 68, 2.0 ** 34), (2, 3, 2, 1234567.0 * 2.0 ** 20, 2.0 ** 40 + 2.0 ** 7),
                                                           (5, 4, 1, 2.0 ** 34, 2.0 ** 34)])
def test_gpu_scaled_refresh_shares_and_finish_bit_exact(level, W, nct, ct_scale, target):
    """the bootstrap the reference calls on its products (levels {4 -> 9}, scale 2^68 -> 2^34) - HIP vs oracle, every word"""
    from sfgwas_amd import capi
    ctx = capi.Context(ol.Q_PN14, ol.P_PN14)
    ring = ol.Ring(14, ol.Q_PN14, ol.P_PN14)
    sk = ol.secret_ntt(ring, ring.gen_secret(4))
    ctx.load_secret_key(sk)
    rnd = np.random.default_rng(300 + level)
    Ql = 1
    for q in ring.moduli[:level + 1]:
        Ql *= q
    bound = Ql // 6
    cts = np.stack([ring.fill_uniform(level, 20 + i) for i in range(nct)])
    crs = np.stack([np.stack([rnd.integers(0, ring.moduli[j], ring.N, dtype=np.uint64) for j in range(ring.nq)]) for _ in range(nct)])
    limbs = np.stack([make_masks(rnd, ring.N, bound, W)[1] for _ in range(nct)])
    ii = int(ct_scale)
    edge = [0, 1, -1, (bound >> 1) - 1, -(bound >> 1), ii, -ii, ii - 1, 1 - ii, ii + 1]
    limbs[0, :len(edge)] = ol.bigints_to_limbs(edge, W)
    e0 = rnd.integers(-19, 20, (nct, ring.N)).astype(np.int32)
    e1 = rnd.integers(-19, 20, (nct, ring.N)).astype(np.int32)
    h0, h1 = ctx.refresh_gen_shares(cts, level, crs, limbs, e0, e1, scales=(ct_scale, target))
    for i in range(nct):
        w0, w1 = ol.refresh_gen_shares_scaled(ring, level, cts[i], ct_scale, target, sk, crs[i], limbs[i], e0[i], e1[i])
        assert np.array_equal(h0[i], w0), f"h0 of ciphertext {i}"
        assert np.array_equal(h1[i], w1), f"h1 of ciphertext {i}"
    h0agg = np.stack([np.stack([rnd.integers(0, ring.moduli[j], ring.N, dtype=np.uint64) for j in range(level + 1)]) for _ in range(nct)])
    h1agg = np.stack([np.stack([rnd.integers(0, ring.moduli[j], ring.N, dtype=np.uint64) for j in range(ring.nq)]) for _ in range(nct)])
    got = ctx.refresh_finish(cts, level, h0agg, h1agg, crs, scales=(ct_scale, target))
    for i in range(nct):
        want = ol.refresh_finish_scaled(ring, level, cts[i], ct_scale, target, h0agg[i], h1agg[i], crs[i])
        assert np.array_equal(got[i], want), f"refreshed ciphertext {i}"
    if ct_scale == target:
        assert np.array_equal(got, ctx.refresh_finish(cts, level, h0agg, h1agg, crs))
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("level,W,nct", [(2, 3, 2), (5, 4, 3), (9, 6, 1)])
def test_gpu_refresh_shares_and_finish_bit_exact(level, W, nct):
    from sfgwas_amd import capi
    ctx = capi.Context(ol.Q_PN14, ol.P_PN14)
    ring = ol.Ring(14, ol.Q_PN14, ol.P_PN14)
    sk = ol.secret_ntt(ring, ring.gen_secret(4))
    ctx.load_secret_key(sk)
    rnd = np.random.default_rng(100 + level)
    Ql = 1
    for q in ring.moduli[:level + 1]:
        Ql *= q
    bound = Ql // 6
    cts = np.stack([ring.fill_uniform(level, 20 + i) for i in range(nct)])
    crs = np.stack([np.stack([rnd.integers(0, ring.moduli[j], ring.N, dtype=np.uint64) for j in range(ring.nq)]) for _ in range(nct)])
    limbs = np.stack([make_masks(rnd, ring.N, bound, W)[1] for _ in range(nct)])
    # edge values in the first coefficients: 0, +-1, the extremes of the mask range
    edge = [0, 1, -1, (bound >> 1) - 1, -(bound >> 1)]
    limbs[0, :len(edge)] = ol.bigints_to_limbs(edge, W)
    e0 = rnd.integers(-19, 20, (nct, ring.N)).astype(np.int32)
    e1 = rnd.integers(-19, 20, (nct, ring.N)).astype(np.int32)
    h0, h1 = ctx.refresh_gen_shares(cts, level, crs, limbs, e0, e1)
    for i in range(nct):
        w0, w1 = ol.refresh_gen_shares(ring, level, cts[i], sk, crs[i], limbs[i], e0[i], e1[i])
        assert np.array_equal(h0[i], w0), f"h0 of ciphertext {i}"
        assert np.array_equal(h1[i], w1), f"h1 of ciphertext {i}"
    # finish with "aggregated" shares = arbitrary residues (the protocol adds the parties' shares modulo q)
    h0agg = np.stack([np.stack([rnd.integers(0, ring.moduli[j], ring.N, dtype=np.uint64) for j in range(level + 1)]) for _ in range(nct)])
    h1agg = np.stack([np.stack([rnd.integers(0, ring.moduli[j], ring.N, dtype=np.uint64) for j in range(ring.nq)]) for _ in range(nct)])
    got = ctx.refresh_finish(cts, level, h0agg, h1agg, crs)
    for i in range(nct):
        want = ol.refresh_finish(ring, level, cts[i], h0agg[i], h1agg[i], crs[i])
        assert np.array_equal(got[i], want), f"refreshed ciphertext {i}"
    ctx.close()


@pytest.mark.gpu
def test_gpu_recode_tie_at_half_modulus():
    """x == floor(Q/2) and its neighbours: lattigo's Cmp(QHalf) in {0, 1} subtracts Q"""
    from sfgwas_amd import capi
    level = 2
    ctx = capi.Context(ol.Q_PN14, ol.P_PN14)
    ring = ol.Ring(14, ol.Q_PN14, ol.P_PN14)
    mods = ring.moduli[:level + 1]
    Q = 1
    for q in mods:
        Q *= q
    H = Q >> 1
    vals = [H, H - 1, H + 1, 0, Q - 1, 1] + [0] * (ring.N - 6)
    ct = np.zeros((1, 2, level + 1, ring.N), dtype=np.uint64)
    for j, q in enumerate(mods):
        ct[0, 0, j] = ring.ntt(j, np.array([v % q for v in vals], dtype=np.uint64))
    zero0 = np.zeros((1, level + 1, ring.N), dtype=np.uint64)
    zero1 = np.zeros((1, ring.nq, ring.N), dtype=np.uint64)
    got = ctx.refresh_finish(ct, level, zero0, zero1, zero1)
    want = ol.refresh_finish(ring, level, ct[0], zero0[0], zero1[0], zero1[0])
    assert np.array_equal(got[0], want)
    signed = [H - Q, H - 1, H + 1 - Q, 0, -1, 1]
    for j in range(ring.nq):
        coeffs = ring.intt(j, got[0, 0, j])
        assert [int(x) for x in coeffs[:6]] == [v % ring.moduli[j] for v in signed]
    ctx.close()


@pytest.mark.gpu
def test_gpu_ckks_to_ss_share_is_the_decrypt_share_and_the_mask_plaintext():
    """MPC.CMatToSS ring work (mpc/ss.go:222-236): h0 equals GenShares' decrypt share; mask_ntt equals NTT(mask) = the same share with sk = 0, e0 = 0"""
    from sfgwas_amd import capi
    level, W, nct = 3, 4, 2
    ctx = capi.Context(ol.Q_PN14, ol.P_PN14)
    ring = ol.Ring(14, ol.Q_PN14, ol.P_PN14)
    sk = ol.secret_ntt(ring, ring.gen_secret(4))
    ctx.load_secret_key(sk)
    rnd = np.random.default_rng(55)
    Ql = 1
    for q in ring.moduli[:level + 1]:
        Ql *= q
    cts = np.stack([ring.fill_uniform(level, 70 + i) for i in range(nct)])
    limbs = np.stack([make_masks(rnd, ring.N, Ql // 4, W)[1] for _ in range(nct)])
    e0 = rnd.integers(-19, 20, (nct, ring.N)).astype(np.int32)
    zero_e, zero_crs, zero_sk = np.zeros(ring.N, dtype=np.int32), np.zeros((ring.nq, ring.N), dtype=np.uint64), np.zeros_like(sk)
    h0, mk = ctx.ckks_to_ss_share(cts, level, limbs, e0)
    for i in range(nct):
        assert np.array_equal(h0[i], ol.refresh_gen_shares(ring, level, cts[i], sk, zero_crs, limbs[i], e0[i], zero_e)[0])
        assert np.array_equal(mk[i], ol.refresh_gen_shares(ring, level, cts[i], zero_sk, zero_crs, limbs[i], zero_e, zero_e)[0])
    ctx.close()
