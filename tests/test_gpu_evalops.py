"""GPU parity of the evaluator operations applied between the matrix products:
MulRelin, Rescale, ct x plaintext, Sub, InnerSumAll - bit-exact against the oracle's lattigo restatement."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
L = ol.lib


@pytest.fixture(scope="module")
def env():
    from sfgwas_amd import capi
    ctx = capi.Context(ol.Q_PN14, ol.P_PN14)
    ring = ol.Ring(14, ol.Q_PN14, ol.P_PN14)
    s = ring.gen_secret(9)
    rlk = np.zeros((ring.beta, 2, len(ring.moduli), ring.N), dtype=np.uint64)
    L().orc_gen_rlk(ring.h, ol.pi8(s), 77, ol.p64(rlk))
    ctx.load_relinkey(rlk)
    yield ctx, ring, s, rlk
    ctx.close()


@pytest.mark.parametrize("level", [9, 5, 2])
def test_mulrelin_and_rescale_bit_exact(env, level):
    ctx, ring, s, rlk = env
    n = 3
    a = np.stack([ring.fill_uniform(level, 10 + j) for j in range(n)])
    b = np.stack([ring.fill_uniform(level, 50 + j) for j in range(n)])
    got = ctx.evalop("sfg_ct_mulrelin_dev", level, a, b)
    for j in range(n):
        want = np.zeros_like(a[j])
        L().orc_mulrelin(ring.h, level, ol.p64(a[j]), ol.p64(b[j]), ol.p64(rlk), ol.p64(want))
        assert np.array_equal(got[j], want), f"mulrelin ct {j} level {level}"
    res = ctx.evalop("sfg_ct_rescale_dev", level, got, out_level=level - 1)
    for j in range(n):
        want = np.zeros((2, level, ring.N), dtype=np.uint64)
        L().orc_rescale(ring.h, level, ol.p64(np.ascontiguousarray(got[j])), ol.p64(want))
        assert np.array_equal(res[j], want), f"rescale ct {j} level {level}"


def test_rescale_at_level_zero_fails_like_lattigo(env):
    ctx, ring, s, rlk = env
    from sfgwas_amd.capi import SfgError
    a = np.stack([ring.fill_uniform(0, 1)])
    with pytest.raises(SfgError, match="already at level 0"):
        ctx.evalop("sfg_ct_rescale_dev", 0, a, out_level=0)


def test_mul_plain_and_sub_bit_exact(env):
    ctx, ring, s, rlk = env
    level, n = 5, 2
    a = np.stack([ring.fill_uniform(level, 3 + j) for j in range(n)])
    b = np.stack([ring.fill_uniform(level, 30 + j) for j in range(n)])
    rnd = np.random.default_rng(3)
    pts = np.stack([ring.encode_ntt((rnd.random(ring.slots) < 0.5).astype(np.float64), 2.0 ** 34, level + 1) for _ in range(n)])
    for stride, label in [((level + 1) * ring.N, "per-ct"), (0, "shared")]:
        got = ctx.evalop("sfg_ct_mul_plain_dev", level, a, pts, extra=(stride,))
        for j in range(n):
            want = np.zeros_like(a[j])
            L().orc_mul_plain(ring.h, level, ol.p64(a[j]), ol.p64(pts[j if stride else 0]), ol.p64(want))
            assert np.array_equal(got[j], want), label
    got = ctx.evalop("sfg_ct_sub_dev", level, a, b)
    for j in range(n):
        want = np.zeros_like(a[j])
        L().orc_ct_addsub(ring.h, level, ol.p64(a[j]), ol.p64(b[j]), 1, ol.p64(want))
        assert np.array_equal(got[j], want)


def test_const_and_plain_linear_ops_bit_exact(env):
    """MultByConst / AddConst / AddNew(ct, pt): device kernels fed with lattigo's scaleUpExact residues vs the oracle"""
    import ctypes as C
    ctx, ring, s, rlk = env
    level, n, scale = 5, 2, 2.0 ** 34
    a = np.stack([ring.fill_uniform(level, 300 + j) for j in range(n)])
    for const in [0.125, -3.0, -1.0 / 8192]:
        sm = C.c_double(); want = np.zeros_like(a[0])
        L().orc_mul_const(ring.h, level, ol.p64(a[0]), const, ol.p64(want), C.byref(sm))
        sc = [L().orc_scale_up_exact(const, sm.value, q) % q for q in ring.moduli[:level + 1]]
        got = ctx.evalop("sfg_ct_mul_scalar_dev", level, a, extra=(sc,))
        assert np.array_equal(got[0], want), const
        L().orc_add_const(ring.h, level, ol.p64(a[1]), const, scale, ol.p64(want))
        sc = [L().orc_scale_up_exact(const, scale, q) % q for q in ring.moduli[:level + 1]]
        got = ctx.evalop("sfg_ct_add_scalar_dev", level, a, extra=(sc,))
        assert np.array_equal(got[1], want), const
    rnd = np.random.default_rng(6)
    pts = np.stack([ring.encode_ntt(rnd.normal(size=ring.slots), scale, level + 1) for _ in range(n)])
    got = ctx.evalop("sfg_ct_add_plain_dev", level, a, pts, extra=((level + 1) * ring.N,))
    for j in range(n):
        want = np.zeros_like(a[j])
        L().orc_add_plain(ring.h, level, ol.p64(a[j]), ol.p64(pts[j]), ol.p64(want))
        assert np.array_equal(got[j], want)


def test_encode_float_vectors_bit_exact(env):
    """crypto.EncodeFloatVector on the device (behind Mask / MaskTrunc / CPMult operands): arbitrary real vectors, MaxLevel"""
    ctx, ring, s, rlk = env
    rnd = np.random.default_rng(8)
    vals = np.stack([rnd.normal(size=ring.slots) * 3.0, (rnd.random(ring.slots) < 0.5).astype(np.float64), np.zeros(ring.slots)])
    vals[2, 17] = 1.0
    got = ctx.encode_vectors(vals, 9)
    for k in range(3):
        assert np.array_equal(got[k], ring.encode_ntt(vals[k], 2.0 ** 34, 10)), f"vector {k}"
    got3 = ctx.encode_vectors(vals[:1], 3)
    assert np.array_equal(got3[0], got[0][:4])                 # a lower level is a prefix of the rows


def test_innersum_all_bit_exact_and_decrypts_to_total(env):
    ctx, ring, s, rlk = env
    import pyref
    level = 3
    keys = ol.RotKeys(ring)
    keys.gen_for_rotations(s, [1 << k for k in range(13)])
    for g, k in keys.keys.items():
        ctx.load_rotkey(g, k)
    rnd = np.random.default_rng(4)
    u, v = rnd.normal(size=ring.slots), rnd.normal(size=ring.slots)
    cts = np.stack([ring.encrypt(s, level, ring.encode_coeffs(x, 2.0 ** 34), 11 + k) for k, x in enumerate((u, v))])
    got = ctx.innersum(cts, level)
    want = np.zeros_like(cts[0])
    assert L().orc_innersum_all(ring.h, keys.h, level, ol.p64(cts), 2, ol.p64(want)) == 0
    assert np.array_equal(got, want)
    res = ring.decrypt_residues(s, level, got)
    big = pyref.crt_centered([res[0], res[1]], ring.moduli[:2])
    slots = pyref.decode(np.array([float(x) for x in big]) / 2.0 ** 34, ring.N)
    assert np.max(np.abs(slots.real - (u.sum() + v.sum()))) < 1e-2


def test_mulrelin_without_key_fails_loudly():
    from sfgwas_amd import capi
    ctx = capi.Context(ol.Q_PN14, ol.P_PN14)
    ring = ol.Ring(14, ol.Q_PN14, ol.P_PN14)
    a = np.stack([ring.fill_uniform(2, 1)])
    with pytest.raises(capi.SfgError, match="no relinearisation key"):
        ctx.evalop("sfg_ct_mulrelin_dev", 2, a, a)
    ctx.close()
