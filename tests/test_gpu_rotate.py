"""GPU parity of rotations (sfg_rotate_right_dev: automorphism + hybrid key switch) against the oracle's
restatement of RotateRightWithEvaluator -> RotateNew (basics.go:201-224). Bit-exact, levels 5 and 4."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from sfgwas_amd import capi
    ctx = capi.Context(ol.Q_PN14, ol.P_PN14)
    ring = ol.Ring(14, ol.Q_PN14, ol.P_PN14)
    s = ring.gen_secret(5)
    keys = ol.RotKeys(ring)
    rots_right = [1, 3, 90, 91 * 7, 8191]
    keys.gen_for_rotations(s, [ring.slots - r for r in rots_right])
    for g, k in keys.keys.items():
        ctx.load_rotkey(g, k)
    yield ctx, ring, keys, rots_right
    ctx.close()


@pytest.mark.parametrize("level", [5, 4, 1])
def test_rotate_right_bit_exact(env, level):
    ctx, ring, keys, rots = env
    nrots = [0] + rots + [rots[0]]
    cts = np.stack([ring.fill_uniform(level, 100 + j) for j in range(len(nrots))])
    got = ctx.rotate_right(cts, level, nrots)
    for j, r in enumerate(nrots):
        want = ol.rotate_right(ring, keys, level, cts[j], r)
        assert np.array_equal(got[j], want), f"ct {j} rot {r} level {level}"


def test_rotate_missing_key_fails_loudly(env):
    ctx, ring, keys, rots = env
    from sfgwas_amd.capi import SfgError
    cts = np.stack([ring.fill_uniform(5, 1)])
    with pytest.raises(SfgError, match="no rotation key"):
        ctx.rotate_right(cts, 5, [2])


def test_montgomery_form_key_upload(env):
    """keys handed over in lattigo's Montgomery representation give the same rotation"""
    ctx, ring, keys, rots = env
    g = ring.galois(ring.slots - rots[1])
    key = keys.keys[g]
    mont = key.copy()
    nmod = len(ring.moduli)
    for m in range(nmod):
        q = ring.moduli[m]
        blk = np.ascontiguousarray(mont[:, :, m, :]).reshape(-1)
        ol.lib().orc_mform_vec(ol.p64(blk), blk.size, q)
        mont[:, :, m, :] = blk.reshape(mont[:, :, m, :].shape)
    ctx.load_rotkey(g, mont, montgomery=True)
    cts = np.stack([ring.fill_uniform(5, 7)])
    got = ctx.rotate_right(cts, 5, [rots[1]])
    assert np.array_equal(got[0], ol.rotate_right(ring, keys, 5, cts[0], rots[1]))
    ctx.load_rotkey(g, key)


def test_conjugate_and_generic_galois_element_bit_exact(env):
    """eval.ConjugateNew (crypto.ComplexConjugate / CReal, basics.go:826-846): automorphism by 2N-1 with its own switching key; a real key from the
    toy secret also shows the semantics: the decrypted coefficient vector is the input's under X -> X^(2N-1)"""
    import ctypes as C
    from sfgwas_amd import capi
    ctx, ring, keys, rots = env
    g = 2 * ring.N - 1
    s = ring.gen_secret(5)
    key = ring.gen_rotkey(s, g, 4242)
    keys.add(g, key)
    ctx.load_rotkey(g, key)
    level = 4
    m = np.random.default_rng(2).integers(-(1 << 20), 1 << 20, ring.N)
    cts = np.stack([ring.encrypt(s, level, m, 31), ring.fill_uniform(level, 32)])
    d_in = ctx.to_device(cts); d_out = ctx.malloc(cts.nbytes)
    ctx.check(capi.lib().sfg_ct_galois_dev(ctx.h, d_in, d_out, 2, level, g), "galois")
    got = ctx.to_host(d_out, cts.shape, np.uint64)
    for j in range(2):
        want = np.zeros_like(cts[j])
        assert ol.lib().orc_apply_galois(ring.h, keys.h, level, ol.p64(cts[j]), g, ol.p64(want)) == 0
        assert np.array_equal(got[j], want), f"ciphertext {j}"
    # semantics: p(X) -> p(X^(2N-1)) = p(X^-1): coefficient c moves to N - c with a sign flip (c > 0)
    q0 = ring.moduli[0]
    dec = ring.decrypt_residues(s, level, got[0])[0]                      # coefficient domain, modulus 0
    dec = np.array([int(v) - q0 if int(v) > q0 // 2 else int(v) for v in dec])
    want_m = np.zeros(ring.N, dtype=np.int64); want_m[0] = m[0]; want_m[1:] = -m[:0:-1]
    assert np.max(np.abs(dec - want_m)) < 1 << 16          # key-switch noise only
    with pytest.raises(capi.SfgError, match="no switching key"):
        ctx.check(capi.lib().sfg_ct_galois_dev(ctx.h, d_in, d_out, 2, level, 7), "galois")
    ctx.free(d_in); ctx.free(d_out)
