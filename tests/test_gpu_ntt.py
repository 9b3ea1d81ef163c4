"""GPU parity: NTT / inverse NTT rows through the C-ABI vs the CPU oracle (bit-exact)."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    from sfgwas_amd import capi
    ctx = capi.Context(ol.Q_PN14, ol.P_PN14)
    ring = ol.Ring(14, ol.Q_PN14, ol.P_PN14)
    yield ctx, ring
    ctx.close()


def test_ntt_rows_bit_exact(env):
    ctx, ring = env
    rnd = np.random.default_rng(1)
    nmod = len(ring.moduli)
    mods = list(range(nmod))
    rows = np.stack([rnd.integers(0, ring.moduli[m], ring.N, dtype=np.uint64) for m in mods])
    # edge values: 0, q-1
    rows[:, 0] = 0
    for m in mods:
        rows[m, 1] = ring.moduli[m] - 1
    got = ctx.ntt_rows(rows, mods)
    for m in mods:
        assert np.array_equal(got[m], ring.ntt(m, rows[m])), f"forward NTT differs for modulus {m}"
    back = ctx.ntt_rows(got, mods, inverse=True)
    assert np.array_equal(back, rows)
    inv = ctx.ntt_rows(rows, mods, inverse=True)
    for m in mods:
        assert np.array_equal(inv[m], ring.intt(m, rows[m])), f"inverse NTT differs for modulus {m}"


def test_ntt_all_max_rows(env):
    ctx, ring = env
    mods = [0, 1, 10, 11]
    rows = np.stack([np.full(ring.N, ring.moduli[m] - 1, dtype=np.uint64) for m in mods])
    got = ctx.ntt_rows(rows, mods)
    inv = ctx.ntt_rows(rows, mods, inverse=True)
    for i, m in enumerate(mods):
        assert np.array_equal(got[i], ring.ntt(m, rows[i]))
        assert np.array_equal(inv[i], ring.intt(m, rows[i]))
