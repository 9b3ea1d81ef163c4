"""GPU parity of the remaining hot-path rows: Beaver local products (B1-B3), count-sketch + moments (P1),
column sums (P2), float-vector encode (Mask path), synthetic generators. All against the oracle / big ints."""
import ctypes as C
import random
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from sfgwas_amd import capi
    c = capi.Context(ol.Q_PN14, ol.P_PN14)
    yield c
    c.close()


def _limbs(x, n):
    return [(x >> (64 * i)) & ((1 << 64) - 1) for i in range(n)]


@pytest.mark.parametrize("limbs,p", [(2, (1 << 127) - 1), (4, (1 << 255) - 19), (2, (1 << 128) - 159), (4, (1 << 256) - 189),           # pseudo-Mersenne: folded reduction
                                     (2, 0xC3A5C85C97CB3127B492B66FBE98F273), (4, 0x9E3779B97F4A7C15F39CC0605CEDC8341082276BF3A27251F86C6A11D0C18E95),  # generic odd moduli: Montgomery path
                                     (2, (1 << 127) - (1 << 40) - 1)])                                                                     # 2^127 - c with c >= 2^32: not the folded form
@pytest.mark.parametrize("pid", [0, 1, 2])
def test_beaver_elem_matches_oracle_and_bigint(ctx, limbs, p, pid):
    from sfgwas_amd import capi
    rnd = random.Random(limbs * 10 + pid)
    n = 3000
    vals = [[rnd.randrange(p) for _ in range(n)] for _ in range(4)]
    for v in vals:                       # edge values
        v[0], v[1], v[2] = 0, p - 1, 1
    arrs = [np.array([_limbs(x, limbs) for x in v], dtype=np.uint64) for v in vals]
    mod = np.array(_limbs(p, limbs), dtype=np.uint64)
    got = np.zeros((n, limbs), dtype=np.uint64)
    ctx.check(capi.lib().sfg_beaver_elem(ctx.h, pid, limbs, capi.p64(mod), *[capi.p64(a) for a in arrs], capi.p64(got), n), "beaver_elem")
    want = np.zeros((n, limbs), dtype=np.uint64)
    ol.lib().orc_beaver_elem(pid, limbs, ol.p64(mod), *[ol.p64(a) for a in arrs], ol.p64(want), n)
    assert np.array_equal(got, want)
    ar, am, br, bm = vals
    for i in range(0, n, 97):            # and directly against Python integers (beavermult.go:94-106)
        e = am[i] * bm[i] % p if pid == 0 else (ar[i] * bm[i] + br[i] * am[i] + (ar[i] * br[i] if pid == 1 else 0)) % p
        assert sum(int(got[i, k]) << (64 * k) for k in range(limbs)) == e


@pytest.mark.parametrize("limbs,p,nparty", [(2, (1 << 127) - 1, 3), (4, (1 << 255) - 19, 3), (4, (1 << 255) - 19, 2), (2, (1 << 127) - 735, 4)])
def test_ss_to_cmat_share_algebra_matches_oracle_and_bigint(ctx, limbs, p, nparty):
    """mpc/ss.go:84-110 (SSToCMat): mask recentring around bound = p / (4 (nParty - 1)), rm - mask, and the hub's revealed + mask - the device kernel vs the
    oracle and vs Python integers, with draws at 0, bound / 2 - 1, bound / 2 and bound - 1"""
    from sfgwas_amd import capi
    rnd = random.Random(limbs * 100 + nparty)
    n = 4000
    bound = p // (4 * (nparty - 1))
    half = bound >> 1
    rm = [rnd.randrange(p) for _ in range(n)]
    tmp = [rnd.randrange(bound) for _ in range(n)]
    tmp[:4] = [0, half - 1, half, bound - 1]
    rm[:4] = [0, p - 1, 5, half]
    A = lambda v: np.array([_limbs(x, limbs) for x in v], dtype=np.uint64)
    mod, bnd = np.array(_limbs(p, limbs), dtype=np.uint64), np.array(_limbs(bound, limbs), dtype=np.uint64)
    d_rm, d_tmp = ctx.to_device(A(rm)), ctx.to_device(A(tmp))
    d_out, d_mask, d_share = ctx.malloc(n * limbs * 8), ctx.malloc(n * limbs * 8), ctx.malloc(n * limbs * 8)
    ctx.check(capi.lib().sfg_ss_mask_dev(ctx.h, limbs, capi.p64(mod), capi.p64(bnd), d_rm, d_tmp, d_out, d_mask, n), "ss_mask")
    got_masked, got_mask = ctx.to_host(d_out, (n, limbs), np.uint64), ctx.to_host(d_mask, (n, limbs), np.uint64)
    w_masked, w_mask = np.zeros((n, limbs), dtype=np.uint64), np.zeros((n, limbs), dtype=np.uint64)
    ol.lib().orc_ss_mask(limbs, ol.p64(mod), ol.p64(bnd), ol.p64(A(rm)), ol.p64(A(tmp)), ol.p64(w_masked), ol.p64(w_mask), n)
    assert np.array_equal(got_masked, w_masked) and np.array_equal(got_mask, w_mask)
    val = lambda row: sum(int(row[k]) << (64 * k) for k in range(limbs))
    for i in list(range(8)) + list(range(8, n, 131)):
        m = (tmp[i] - bound) % p if tmp[i] >= half else tmp[i]
        assert val(got_mask[i]) == m and val(got_masked[i]) == (rm[i] - m) % p
    # hub: share = revealed + mask; with revealed = rm - mask the hub's share is rm again
    ctx.check(capi.lib().sfg_ss_hub_share_dev(ctx.h, limbs, capi.p64(mod), d_out, d_mask, d_share, n), "ss_hub_share")
    share = ctx.to_host(d_share, (n, limbs), np.uint64)
    assert np.array_equal(share, A(rm))
    w_share = np.zeros((n, limbs), dtype=np.uint64)
    ol.lib().orc_ss_hub_share(limbs, ol.p64(mod), ol.p64(w_masked), ol.p64(w_mask), ol.p64(w_share), n)
    assert np.array_equal(share, w_share)
    for q in (d_rm, d_tmp, d_out, d_mask, d_share):
        ctx.free(q)


@pytest.mark.parametrize("pid", [0, 1, 2])
def test_beaver_matmul(ctx, pid):
    from sfgwas_amd import capi
    limbs, p = 4, (1 << 255) - 19
    rnd = random.Random(pid)
    m, k, n = 15, 15, 7
    mk = lambda cnt: np.array([_limbs(rnd.randrange(p), limbs) for _ in range(cnt)], dtype=np.uint64)
    ar, am, br, bm = mk(m * k), mk(m * k), mk(k * n), mk(k * n)
    mod = np.array(_limbs(p, limbs), dtype=np.uint64)
    got = np.zeros((m * n, limbs), dtype=np.uint64)
    ctx.check(capi.lib().sfg_beaver_matmul(ctx.h, pid, limbs, capi.p64(mod), capi.p64(ar), capi.p64(am), capi.p64(br), capi.p64(bm), capi.p64(got), m, k, n), "beaver_matmul")
    want = np.zeros((m * n, limbs), dtype=np.uint64)
    ol.lib().orc_beaver_matmul(pid, limbs, ol.p64(mod), ol.p64(ar), ol.p64(am), ol.p64(br), ol.p64(bm), ol.p64(want), m, k, n)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("nrow,ncol,full_range", [(1000, 777, False), (64, 256, False), (130, 5, False), (1, 1, False), (1037, 777, True), (200, 300, True)])
def test_sketch_and_moments(ctx, nrow, ncol, full_range):
    """full_range: arbitrary int8 values - squares wrap in int8 (Go: uint64(row[j]*row[j])) and sums go negative: the per-byte path of the moments and
    the sign handling of the int8 matrix-core projection"""
    from sfgwas_amd import capi
    rnd = np.random.default_rng(nrow + ncol)
    X = (rnd.integers(-128, 128, (nrow, ncol)) if full_range else rnd.integers(-1, 3, (nrow, ncol))).astype(np.int8)
    kp = 15
    bucket = rnd.integers(0, kp, nrow).astype(np.int32)
    sgn = (rnd.integers(0, 2, nrow) * 2 - 1).astype(np.int8)
    gh = C.c_void_p()
    ctx.check(capi.lib().sfg_geno_upload(ctx.h, X.ctypes.data_as(C.c_void_p), nrow, ncol, ncol, C.byref(gh)), "geno_upload")
    sk = np.zeros((kp, ncol)); xs = np.zeros(ncol, dtype=np.uint64); x2 = np.zeros(ncol, dtype=np.uint64)
    ctx.check(capi.lib().sfg_sketch(ctx.h, gh, bucket.ctypes.data_as(C.POINTER(C.c_int32)), sgn.ctypes.data_as(C.POINTER(C.c_int8)), kp,
                                    sk.ctypes.data_as(C.POINTER(C.c_double)), capi.p64(xs), capi.p64(x2)), "sketch")
    wsk = np.zeros((kp, ncol)); wxs = np.zeros(ncol, dtype=np.uint64); wx2 = np.zeros(ncol, dtype=np.uint64)
    ol.lib().orc_sketch(ol.pi8(X), nrow, ncol, bucket.ctypes.data_as(C.POINTER(C.c_int32)), ol.pi8(sgn), kp, ol.pd(wsk), ol.p64(wxs), ol.p64(wx2))
    assert np.array_equal(sk, wsk) and np.array_equal(xs, wxs) and np.array_equal(x2, wx2)
    # P2 column sums (missing -> 0)
    sm = np.zeros(ncol); sq = np.zeros(ncol)
    ctx.check(capi.lib().sfg_geno_colsums(ctx.h, gh, sm.ctypes.data_as(C.POINTER(C.c_double)), sq.ctypes.data_as(C.POINTER(C.c_double))), "colsums")
    g0 = np.where(X < 0, 0, X).astype(np.int8)
    with np.errstate(over="ignore"):
        sq0 = (g0 * g0).astype(np.int8)                                   # the reference squares in int8 (wraps from 12 upwards) before float64()
    assert np.array_equal(sm, g0.astype(np.float64).sum(0)) and np.array_equal(sq, sq0.astype(np.float64).sum(0))
    capi.lib().sfg_geno_free(ctx.h, gh)


def test_encode_float_vectors_match_oracle(ctx):
    from sfgwas_amd import capi
    ring = ol.Ring(14, ol.Q_PN14[:2], ol.P_PN14)
    rnd = np.random.default_rng(3)
    vecs = np.stack([np.r_[np.ones(100), np.zeros(8192 - 100)],          # MaskTrunc-style 0/1 mask (basics.go:110-127)
                     rnd.normal(size=8192), np.zeros(8192)])
    out = np.zeros((3, 16384), dtype=np.int64)
    ctx.check(capi.lib().sfg_encode_coeffs_host(ctx.h, vecs.ctypes.data_as(C.POINTER(C.c_double)), 3, out.ctypes.data_as(C.POINTER(C.c_int64))), "encode_coeffs")
    for k in range(3):
        assert np.array_equal(out[k], ring.encode_coeffs(vecs[k], 2.0 ** 34, prec=0))


def test_device_generators_match_host_definitions(ctx):
    from sfgwas_amd import capi
    ring = ol.Ring(14, ol.Q_PN14, ol.P_PN14)
    level, nct = 5, 3
    d = ctx.malloc(nct * 2 * (level + 1) * ring.N * 8)
    ctx.check(capi.lib().sfg_fill_uniform_ct_dev(ctx.h, d, nct, level, 4242), "fill_uniform")
    got = ctx.to_host(d, (nct, 2, level + 1, ring.N), np.uint64)
    ctx.free(d)
    for j in range(nct):
        assert np.array_equal(got[j], ring.fill_uniform(level, 4242 + j))
    nrow, ncol = 300, 1000
    g = ctx.malloc(nrow * ncol)
    ctx.check(capi.lib().sfg_fill_geno_dev(ctx.h, g, nrow, ncol, 99), "fill_geno")
    X = ctx.to_host(g, (nrow, ncol), np.int8)
    ctx.free(g)
    assert set(np.unique(X)) <= {-1, 0, 1, 2}
    assert 0.003 < (X == -1).mean() < 0.02                     # ~1/128 missing
    maf = np.where(X < 0, 0, X).mean(0) / 2
    assert 0.03 < maf.min() and maf.max() < 0.6                # p_j ~ U(0.05, 0.5)


def test_public_configuration_struct_changes_schedules_not_words():
    """sfg_config (include/sfgwas_hip.h) through sfg_ctx_create_ex: MAC group size, accumulator budget, encode batch - the deployment's knobs - give the words of the
    default context; a struct from an OLDER header (smaller struct_size: trailing fields unknown to the caller) is accepted; sfg_config_default zeroes and sizes it."""
    import ctypes as C
    from sfgwas_amd import capi
    import oracle_lib as ol
    lib = capi.lib()
    cfg = capi.SfgConfig()
    cfg.mm_group = 99
    lib.sfg_config_default(C.byref(cfg))
    assert cfg.struct_size == C.sizeof(capi.SfgConfig) and cfg.mm_group == 0
    D, LEVEL, L = 91, 5, 5
    rots = list(range(1, D)) + [g * D for g in range(1, D) if g * D < 8192]
    outs = []
    old = capi.SfgConfig(mm_group=1, enc_batch=300)
    old.struct_size = capi.SfgConfig.acc_budget_bytes.offset                      # a caller compiled against a header that ended after mm_group
    for config in (None, capi.SfgConfig(mm_group=1, acc_budget_bytes=300 << 20, enc_batch=512, upload_blocking=1), old):
        ctx = capi.Context(ol.Q_PN14, ol.P_PN14, config=config)
        ctx.check(lib.sfg_fill_rotkeys_synthetic(ctx.h, (C.c_int * len(rots))(*rots), len(rots), 0xBEEF), "keys")
        nrow, ncol, s = 2 * 8192 + 40, 8192 + 9, 2
        gd, gh = ctx.fill_geno(nrow, ncol, 78)
        A = ctx.fill_uniform_cts(s * 3, LEVEL, 6)
        out = ctx.matmul_resident(A, s, LEVEL, L, gh)
        outs.append(out.host().copy())
        for d in (A, out, gd):
            d.free()
        ctx.geno_free(gh); ctx.close()
    assert outs[0].any() and np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2])
