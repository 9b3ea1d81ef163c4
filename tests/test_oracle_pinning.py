"""Pins the CPU oracle (oracle/sfgwas_oracle.c) against independent big-integer / mpmath statements.

The reference ships no tests or golden vectors, so these known-answer and
identity tests are what stands behind every parity claim of the HIP path.
"""
import ctypes as C
import os
import random
import numpy as np
import pytest

import oracle_lib as ol
import pyref

L = ol.lib


def small_ring(logN=5, nq=6, np_=2):
    q = ol.small_primes(logN, 46, 1) + ol.small_primes(logN, 35, nq - 1)
    p = ol.small_primes(logN, 43, np_)
    return ol.Ring(logN, q, p)


# ---------------------------------------------------------------- integer kernels specified in matmult.go
def test_mred_bred_mform_params():
    rnd = random.Random(1)
    for q in ol.Q_PN14 + ol.P_PN14:
        qinv = L().orc_mred_params(q)
        assert (qinv * q) % (1 << 64) == 1
        u = np.zeros(2, dtype=np.uint64)
        L().orc_bred_params(q, ol.p64(u))
        assert (int(u[0]) << 64) + int(u[1]) == (1 << 128) // q
        for _ in range(200):
            a = rnd.randrange(q)
            assert L().orc_mform(a, q, ol.p64(u)) == (a << 64) % q          # matmult.go:433-440
            b = rnd.randrange(q)
            assert L().orc_mred(a, b, q, qinv) == a * b * pow(1 << 64, -1, q) % q


@pytest.mark.parametrize("q", [ol.Q_PN14[0], ol.Q_PN14[1], ol.Q_PN14[2]])
def test_lazy_mac_redc_is_canonical_sum(q):
    """A1+A4+A5: sum_k ct_k * MForm(pt_k) accumulated in u128, REDC'd and reduced == sum ct*pt mod q."""
    rnd = np.random.default_rng(5)
    n, K = 64, 300
    acc = np.zeros((n, 2), dtype=np.uint64)
    expect = [0] * n
    for _ in range(K):
        a = rnd.integers(0, q, n, dtype=np.uint64)
        b = rnd.integers(0, q, n, dtype=np.uint64)
        bm = b.copy()
        L().orc_mform_vec(ol.p64(bm), n, q)
        L().orc_mul_coeffs_and_add128(ol.p64(a), ol.p64(bm), ol.p64(acc), n)
        for j in range(n):
            expect[j] += int(a[j]) * int(b[j])
    # u128 accumulator holds the exact integer sum of ct * pt_mont
    out = np.zeros(n, dtype=np.uint64)
    L().orc_reduce_and_add_uint128(ol.p64(acc), ol.p64(out), L().orc_mred_params(q), q, n)
    L().orc_canonical_reduce(ol.p64(out), n, q)
    assert [int(x) for x in out] == [e % q for e in expect]


def test_mac_u128_exact_extremes():
    q = (1 << 64) - 59
    a = np.array([q - 1, 1, 0, (1 << 63)], dtype=np.uint64)
    b = np.array([q - 1, q - 1, 5, 2], dtype=np.uint64)
    acc = np.zeros((4, 2), dtype=np.uint64)
    for _ in range(3):
        L().orc_mul_coeffs_and_add128(ol.p64(a), ol.p64(b), ol.p64(acc), 4)
    for j in range(4):
        v = (int(acc[j, 0]) << 64) + int(acc[j, 1])     # {hi, lo} order, matmult.go:196-199
        assert v == (3 * int(a[j]) * int(b[j])) % (1 << 128)


# ---------------------------------------------------------------- NTT layout
def test_ntt_matches_direct_evaluation():
    ring = small_ring(5)
    rnd = np.random.default_rng(2)
    for m in range(len(ring.moduli)):
        q = ring.moduli[m]
        p = rnd.integers(0, q, ring.N, dtype=np.uint64)
        got = ring.ntt(m, p)
        assert [int(x) for x in got] == pyref.ntt_direct(p, ring.psi[m], q, ring.logN)
        assert np.array_equal(ring.intt(m, got), p)


def test_ntt_pointwise_is_negacyclic_convolution():
    ring = small_ring(5)
    rnd = np.random.default_rng(3)
    q = ring.moduli[1]
    a = rnd.integers(0, q, ring.N, dtype=np.uint64)
    b = rnd.integers(0, q, ring.N, dtype=np.uint64)
    fa, fb = ring.ntt(1, a), ring.ntt(1, b)
    prod = np.array([int(x) * int(y) % q for x, y in zip(fa, fb)], dtype=np.uint64)
    assert [int(x) for x in ring.intt(1, prod)] == pyref.negacyclic_mul(a, b, q)


def test_psi_is_from_smallest_primitive_root():
    from sympy.ntheory import primitive_root
    ring = small_ring(5)
    for m, q in enumerate(ring.moduli):
        g = primitive_root(q)
        assert ring.psi[m] == pow(g, (q - 1) // (2 * ring.N), q)


def test_automorphism_index_matches_coefficient_map():
    ring = small_ring(5)
    rnd = np.random.default_rng(4)
    q = ring.moduli[0]
    p = rnd.integers(0, q, ring.N, dtype=np.uint64)
    for k in [1, 3, 7]:
        g = ring.galois(k)
        assert g == pow(5, k, 2 * ring.N)
        idx = np.zeros(ring.N, dtype=np.uint32)
        L().orc_automorphism_index(ring.h, g, idx.ctypes.data_as(C.POINTER(C.c_uint32)))
        lhs = ring.ntt(0, p)[idx]
        rhs = ring.ntt(0, np.array(pyref.automorphism_coeffs(p, g, q), dtype=np.uint64))
        assert np.array_equal(lhs, rhs)


# ---------------------------------------------------------------- encode
@pytest.mark.parametrize("logN", [4, 6])
def test_encode_matches_mpmath_exact_rounding(logN):
    ring = small_ring(logN)
    rnd = np.random.default_rng(7)
    for trial in range(3):
        v = rnd.integers(0, 5, ring.slots).astype(np.float64)
        want = pyref.encode_exact(v, ring.N, 2.0 ** 34)
        assert list(ring.encode_coeffs(v, 2.0 ** 34, prec=0)) == want
        assert list(ring.encode_coeffs(v, 2.0 ** 34, prec=1)) == want


def test_encode_decode_roundtrip_and_real_symmetry_N16384():
    ring = ol.Ring(14, ol.Q_PN14[:6], ol.P_PN14)
    rnd = np.random.default_rng(8)
    v = rnd.integers(0, 3, ring.slots).astype(np.float64)
    c0 = ring.encode_coeffs(v, 2.0 ** 34, prec=0)
    c1 = ring.encode_coeffs(v, 2.0 ** 34, prec=1)
    assert np.array_equal(c0, c1)                     # double-double agrees with 113-bit arithmetic
    back = pyref.decode(c0.astype(np.float64) / 2.0 ** 34, ring.N)
    assert np.max(np.abs(back - v)) < 1e-5
    # real slot vectors give p_{N-c} = -p_c (invariance under X -> X^-1)
    assert c0[ring.N // 2] == 0
    assert np.array_equal(c0[1:], -c0[:0:-1])
    # encode_ntt = NTT of the reduced coefficients
    pt = ring.encode_ntt(v, 2.0 ** 34, 2, prec=1)
    for l in range(2):
        q = ring.moduli[l]
        red = np.array([int(x) % q for x in c0], dtype=np.uint64)
        assert np.array_equal(pt[l], ring.ntt(l, red))


# ---------------------------------------------------------------- diagonals
def test_get_diag_against_bruteforce_and_existence_rule():
    rnd = np.random.default_rng(9)
    dim = 16
    for (r, c) in [(16, 16), (5, 16), (16, 3), (4, 6), (1, 1), (9, 9)]:
        X = rnd.integers(-1, 3, (r, c)).astype(np.int8)
        for shift in range(dim):
            dst = np.full(dim, 7.0)
            ok = L().orc_get_diag(ol.pd(dst), ol.pi8(np.ascontiguousarray(X)), c, r, c, dim, -shift)
            want = pyref.get_diag_bruteforce(X, dim, shift)
            idx = (-shift) % dim
            exists = (dim + 1 - r) <= idx or idx <= c - 1           # matmult.go:627-631
            assert bool(ok) == exists == bool(L().orc_get_diag_bool(r, c, dim, -shift))
            if ok:
                assert np.array_equal(dst, want)
            else:
                assert not want.any()                               # a missing diagonal is all-zero


# ---------------------------------------------------------------- rotation / key switch
def _encode_encrypt(ring, s, level, v, scale, seed):
    return ring.encrypt(s, level, ring.encode_coeffs(v, scale), seed)


def _decrypt_decode(ring, s, level, ct, scale, nmod=2):
    res = ring.decrypt_residues(s, level, ct)
    big = pyref.crt_centered([res[m] for m in range(nmod)], ring.moduli[:nmod])
    return pyref.decode(np.array([float(x) for x in big]) / scale, ring.N)


@pytest.mark.parametrize("level", [5, 4, 2])
def test_rotation_decrypts_to_rotated_vector(level):
    ring = small_ring(6)
    s = ring.gen_secret(11)
    keys = ol.RotKeys(ring)
    keys.gen_for_rotations(s, [1, 5, ring.slots - 3])
    rnd = np.random.default_rng(12)
    v = rnd.normal(size=ring.slots)
    scale = 2.0 ** 34
    ct = _encode_encrypt(ring, s, level, v, scale, 99)
    for k in [1, 5]:
        out = ol.rotate_left(ring, keys, level, ct, k)
        got = _decrypt_decode(ring, s, level, out, scale)
        assert np.max(np.abs(got - np.roll(v, -k))) < 1e-4
    out = ol.rotate_right(ring, keys, level, ct, 3)                  # basics.go:201-210
    got = _decrypt_decode(ring, s, level, out, scale)
    assert np.max(np.abs(got - np.roll(v, 3))) < 1e-4
    same = ol.rotate_right(ring, keys, level, ct, 0)
    assert np.array_equal(same, ct)


# ---------------------------------------------------------------- full matmul
@pytest.mark.parametrize("nrow,ncol,square", [(16, 16, False), (21, 40, False), (37, 9, True), (5, 3, False)])
def test_matmult4stream_decrypts_to_plain_product(nrow, ncol, square):
    ring = small_ring(5)                    # N=32, slots=16, d=4
    slots = ring.slots
    s_ = ring.gen_secret(21)
    rots, d = ol.needed_rotations(slots, nrow, ncol)
    keys = ol.RotKeys(ring)
    keys.gen_for_rotations(s_, rots)
    rnd = np.random.default_rng(nrow * 100 + ncol)
    geno = rnd.integers(-1, 3, (nrow, ncol)).astype(np.int8)
    s, in_level, max_level = 2, 5, 5
    nbr, m_ct = (nrow - 1) // slots + 1, (ncol - 1) // slots + 1
    a_plain = rnd.normal(size=(s, nbr * slots))
    a_plain[:, nrow:] = 0
    scale = 2.0 ** 34
    A = np.zeros((s, nbr, 2, in_level + 1, ring.N), dtype=np.uint64)
    for i in range(s):
        for b in range(nbr):
            A[i, b] = _encode_encrypt(ring, s_, in_level, a_plain[i, b * slots:(b + 1) * slots], scale, 1000 + i * 10 + b)
    out, sm, sq = ol.matmult4stream(ring, keys, scale, A, in_level, max_level, geno, compute_sqsum=True, square=square)
    g0 = np.where(geno < 0, 0, geno).astype(np.float64)
    assert np.array_equal(sm, g0.sum(0)) and np.array_equal(sq, (g0 * g0).sum(0))    # matmult.go:1292-1300
    gx = g0 * g0 if square else g0
    want = a_plain[:, :nrow] @ gx
    for i in range(s):
        for j in range(m_ct):
            got = _decrypt_decode(ring, s_, max_level - 1, out[i, j], scale * scale, nmod=3)
            w = np.zeros(slots)
            seg = want[i, j * slots:(j + 1) * slots]
            w[:len(seg)] = seg
            assert np.max(np.abs(got.real - w)) < 1e-3, (i, j)


# ---------------------------------------------------------------- DiagCache format
def test_diagcache_roundtrip_and_layout(tmp_path):
    d, n, nmod, vlen = 4, 8, 3, 3
    path = str(tmp_path / "cache_0.bin").encode()
    dc = L().orc_diagcache_create(path, d)
    baby = np.array([1, 0, 1, 1], dtype=np.uint8)
    giant = np.array([1, 1, 0, 0], dtype=np.uint8)
    L().orc_diagcache_set_tables(dc, baby.ctypes.data_as(C.POINTER(C.c_uint8)), giant.ctypes.data_as(C.POINTER(C.c_uint8)))
    rnd = np.random.default_rng(1)
    recs = []
    for shift in [0, 2, 7]:
        pv = [rnd.integers(0, 1 << 40, (nmod, n), dtype=np.uint64) if k != 1 or shift != 2 else None for k in range(vlen)]
        arr = (ol.u64p * vlen)(*[ol.p64(p) if p is not None else None for p in pv])
        L().orc_diagcache_write(dc, arr, vlen, 5, 2.0 ** 34, n, nmod, shift)
        recs.append((shift, pv))
    L().orc_diagcache_close(dc)
    raw = open(path.decode(), "rb").read()
    hdr = np.frombuffer(raw[:48], dtype="<u8")
    rowsize = 4 + (1 + n * nmod * 8) * vlen
    assert list(hdr[[0, 1, 3, 4, 5]]) == [vlen, 5, n, nmod, rowsize]                 # filestream.go:148-161
    assert np.frombuffer(raw[16:24], dtype="<f8")[0] == 2.0 ** 34
    assert raw[48:48 + d] == bytes(baby) and raw[48 + d:48 + 2 * d] == bytes(giant)  # :172-187
    first_len = int(np.frombuffer(raw[48 + 2 * d:56 + 2 * d], dtype="<u8")[0])
    assert first_len == rowsize
    dc = L().orc_diagcache_open(path, d)
    bufs = [np.zeros((nmod, n), dtype=np.uint64) for _ in range(vlen)]
    arr = (ol.u64p * vlen)(*[ol.p64(b) for b in bufs])
    empty = np.zeros(vlen, dtype=np.uint8)
    shift = C.c_uint32()
    for want_shift, pv in recs:
        assert L().orc_diagcache_read(dc, arr, empty.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(shift)) == 1
        assert shift.value == want_shift
        for k in range(vlen):
            assert bool(empty[k]) == (pv[k] is None)
            if pv[k] is not None:
                assert np.array_equal(bufs[k], pv[k])
    assert L().orc_diagcache_read(dc, arr, empty.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(shift)) == 0
    L().orc_diagcache_close(dc)


# ---------------------------------------------------------------- Beaver
def _to_limbs(x, limbs):
    return [(x >> (64 * i)) & ((1 << 64) - 1) for i in range(limbs)]


@pytest.mark.parametrize("limbs,p", [(2, (1 << 127) - 1), (4, (1 << 255) - 19)])
def test_beaver_local_products_reconstruct_product(limbs, p):
    rnd = random.Random(3)
    n, nparties = 50, 3
    a = [rnd.randrange(p) for _ in range(n)]
    b = [rnd.randrange(p) for _ in range(n)]
    # additive shares among parties 1..2; Beaver masks am_p, bm_p known to party 0 as sums
    def shares(x):
        s1 = [rnd.randrange(p) for _ in x]
        return [None, s1, [(xi - si) % p for xi, si in zip(x, s1)]]
    am = [None] + [[rnd.randrange(p) for _ in range(n)] for _ in range(nparties - 1)]
    bm = [None] + [[rnd.randrange(p) for _ in range(n)] for _ in range(nparties - 1)]
    am[0] = [(am[1][i] + am[2][i]) % p for i in range(n)]
    bm[0] = [(bm[1][i] + bm[2][i]) % p for i in range(n)]
    ar = [(a[i] - am[0][i]) % p for i in range(n)]          # revealed a - mask (beavermult.go:51-54)
    br = [(b[i] - bm[0][i]) % p for i in range(n)]
    mod = np.array(_to_limbs(p, limbs), dtype=np.uint64)

    def arr(xs):
        return np.array([_to_limbs(x, limbs) for x in xs], dtype=np.uint64)

    outs = []
    for pid in range(nparties):
        out = np.zeros((n, limbs), dtype=np.uint64)
        L().orc_beaver_elem(pid, limbs, ol.p64(mod), ol.p64(arr(ar)), ol.p64(arr(am[pid])), ol.p64(arr(br)), ol.p64(arr(bm[pid])), ol.p64(out), n)
        outs.append([sum(int(out[i, k]) << (64 * k) for k in range(limbs)) for i in range(n)])
    for i in range(n):
        assert outs[0][i] == am[0][i] * bm[0][i] % p                                   # beavermult.go:116-120
        assert outs[1][i] == (ar[i] * bm[1][i] + br[i] * am[1][i] + ar[i] * br[i]) % p  # :125-129
        assert outs[2][i] == (ar[i] * bm[2][i] + br[i] * am[2][i]) % p
        # shares of parties 1..2 plus party 0's am*bm (re-shared by BeaverReconstruct) sum to a*b
        assert (outs[0][i] + outs[1][i] + outs[2][i]) % p == a[i] * b[i] % p


def test_beaver_matmul():
    limbs, p = 2, (1 << 127) - 1
    rnd = random.Random(4)
    m, k, n = 3, 4, 2
    mk = lambda r, c: [[rnd.randrange(p) for _ in range(c)] for _ in range(r)]
    ar, am, br, bm = mk(m, k), mk(m, k), mk(k, n), mk(k, n)
    mod = np.array(_to_limbs(p, limbs), dtype=np.uint64)
    flat = lambda M: np.array([_to_limbs(x, limbs) for row in M for x in row], dtype=np.uint64)
    mm = lambda A, B: [[sum(A[i][x] * B[x][j] for x in range(k)) % p for j in range(n)] for i in range(m)]
    for pid in range(3):
        out = np.zeros((m * n, limbs), dtype=np.uint64)
        L().orc_beaver_matmul(pid, limbs, ol.p64(mod), ol.p64(flat(ar)), ol.p64(flat(am)), ol.p64(flat(br)), ol.p64(flat(bm)), ol.p64(out), m, k, n)
        got = [sum(int(out[i, t]) << (64 * t) for t in range(limbs)) for i in range(m * n)]
        if pid == 0:
            want = mm(am, bm)
        else:
            x, y = mm(ar, bm), mm(am, br)
            want = [[(x[i][j] + y[i][j]) % p for j in range(n)] for i in range(m)]
            if pid == 1:
                z = mm(ar, br)
                want = [[(want[i][j] + z[i][j]) % p for j in range(n)] for i in range(m)]
        assert got == [v for row in want for v in row]


# ---------------------------------------------------------------- sketch
def test_sketch_and_moments():
    rnd = np.random.default_rng(6)
    nrow, ncol, kp = 40, 23, 5
    X = rnd.integers(0, 3, (nrow, ncol)).astype(np.int8)
    bucket = rnd.integers(0, kp, nrow).astype(np.int32)
    sgn = (rnd.integers(0, 2, nrow) * 2 - 1).astype(np.int8)
    sk = np.zeros((kp, ncol))
    xs = np.zeros(ncol, dtype=np.uint64)
    x2 = np.zeros(ncol, dtype=np.uint64)
    L().orc_sketch(ol.pi8(X), nrow, ncol, bucket.ctypes.data_as(C.POINTER(C.c_int32)), ol.pi8(sgn), kp, ol.pd(sk), ol.p64(xs), ol.p64(x2))
    want = np.zeros((kp, ncol))
    for i in range(nrow):
        want[bucket[i]] += sgn[i] * X[i].astype(np.float64)          # pca.go:156
    assert np.array_equal(sk, want)
    assert np.array_equal(xs, X.astype(np.int64).sum(0).astype(np.uint64))
    assert np.array_equal(x2, (X.astype(np.int64) ** 2).sum(0).astype(np.uint64))


# ---------------------------------------------------------------- evaluator ops between the matmuls (C2-C4)
def test_mulrelin_rescale_mulplain_innersum_decrypt_correctly():
    ring = small_ring(6)
    s = ring.gen_secret(31)
    rlk = np.zeros((ring.beta, 2, len(ring.moduli), ring.N), dtype=np.uint64)
    L().orc_gen_rlk(ring.h, ol.pi8(s), 5, ol.p64(rlk))
    keys = ol.RotKeys(ring)
    keys.gen_for_rotations(s, [1 << k for k in range(ring.logN - 1)])
    rnd = np.random.default_rng(13)
    level, scale = 5, 2.0 ** 34
    u, v = rnd.normal(size=ring.slots), rnd.normal(size=ring.slots)
    cu, cv = _encode_encrypt(ring, s, level, u, scale, 1), _encode_encrypt(ring, s, level, v, scale, 2)
    # MulRelin + one Rescale step (CMult, basics.go:386-427)
    prod = np.zeros_like(cu)
    L().orc_mulrelin(ring.h, level, ol.p64(cu), ol.p64(cv), ol.p64(rlk), ol.p64(prod))
    got = _decrypt_decode(ring, s, level, prod, scale * scale, nmod=3)
    assert np.max(np.abs(got.real - u * v)) < 1e-3
    res = np.zeros((2, level, ring.N), dtype=np.uint64)
    L().orc_rescale(ring.h, level, ol.p64(prod), ol.p64(res))
    got = _decrypt_decode(ring, s, level - 1, res, scale * scale / ring.moduli[level], nmod=2)
    assert np.max(np.abs(got.real - u * v)) < 1e-3
    # ct x plaintext (Mask path, basics.go:110-172): 0/1 mask
    mask = (np.arange(ring.slots) < 5).astype(np.float64)
    pt = ring.encode_ntt(mask, scale, level + 1)
    mp = np.zeros_like(cu)
    L().orc_mul_plain(ring.h, level, ol.p64(cu), ol.p64(pt), ol.p64(mp))
    got = _decrypt_decode(ring, s, level, mp, scale * scale, nmod=3)
    assert np.max(np.abs(got.real - u * mask)) < 1e-3
    # Sub and InnerSumAll (basics.go:278-292): every slot holds the total
    df = np.zeros_like(cu)
    L().orc_ct_addsub(ring.h, level, ol.p64(cu), ol.p64(cv), 1, ol.p64(df))
    got = _decrypt_decode(ring, s, level, df, scale)
    assert np.max(np.abs(got.real - (u - v))) < 1e-4
    both = np.ascontiguousarray(np.stack([cu, cv]))
    tot = np.zeros_like(cu)
    assert L().orc_innersum_all(ring.h, keys.h, level, ol.p64(both), 2, ol.p64(tot)) == 0
    got = _decrypt_decode(ring, s, level, tot, scale)
    assert np.max(np.abs(got.real - (u.sum() + v.sum()))) < 1e-3


def test_mul_const_add_const_add_plain_decrypt_correctly():
    """MultByConst / AddConst / AddNew(ct, pt) restated (basics.go:183-199, 480-497, 604-611): decrypted semantics and the
    scaleUpExact rule against Python integers"""
    ring = small_ring(6)
    s = ring.gen_secret(3)
    rnd = np.random.default_rng(17)
    level, scale = 4, 2.0 ** 34
    u = rnd.normal(size=ring.slots)
    cu = _encode_encrypt(ring, s, level, u, scale, 5)
    for const in [3.0, -2.0, 0.125, -1.0 / 8192, 0.0]:
        out = np.zeros_like(cu); sm = C.c_double()
        L().orc_mul_const(ring.h, level, ol.p64(cu), const, ol.p64(out), C.byref(sm))
        frac = const != int(const)
        assert sm.value == (float(ring.moduli[level]) if frac else 1.0)
        got = _decrypt_decode(ring, s, level, out, scale * sm.value, nmod=3)
        assert np.max(np.abs(got.real - const * u)) < 1e-3
        out = np.zeros_like(cu)
        L().orc_add_const(ring.h, level, ol.p64(cu), const, scale, ol.p64(out))
        got = _decrypt_decode(ring, s, level, out, scale)
        assert np.max(np.abs(got.real - (u + const))) < 1e-4
    for q in ring.moduli[:3]:
        for v, n in [(0.3, 2.0 ** 34), (-0.3, 2.0 ** 34), (7.0, 1.0), (-7.0, 1.0), (1 / 8192, float(q)), (-1e-30, 1.0)]:
            want = int(abs(n * v) + 0.5) % q
            want = (q - want) % q if v < 0 else want
            assert L().orc_scale_up_exact(v, n, q) % q == want
    v = rnd.normal(size=ring.slots)
    pt = ring.encode_ntt(v, scale, level + 1)
    out = np.zeros_like(cu)
    L().orc_add_plain(ring.h, level, ol.p64(cu), ol.p64(pt), ol.p64(out))
    got = _decrypt_decode(ring, s, level, out, scale)
    assert np.max(np.abs(got.real - (u + v))) < 1e-4


def test_mul_const_and_add_semantics_on_decrypted_values():
    """eval.MultByConstAndAdd restatement (parity unpinned): whatever branch of the scale matching runs, the receiver afterwards decrypts to
    out/scale_out + constant * in/scale_in (up to the rounding of the scaled constant and the truncated integer ratio)"""
    import ctypes as C
    ring = small_ring(4)
    s = ring.gen_secret(3)
    rnd = np.random.default_rng(9)
    SC = 2.0 ** 30
    for const, s0, so in [(-3.0, SC, SC), (-2.0 / 3000.0, SC, SC), (5.0, SC, SC * 64.0), (-7.0, SC * 1000.0, SC), (0.375, SC, SC * 2.0 ** 20)]:
        level = 2
        va, vo = rnd.uniform(-1, 1, ring.N), rnd.uniform(-1, 1, ring.N)
        a = ring.encrypt(s, level, np.round(va * s0).astype(np.int64), 5)
        o = ring.encrypt(s, level, np.round(vo * so).astype(np.int64), 6)
        sc = C.c_double(so)
        ol.lib().orc_mul_const_and_add(ring.h, level, ol.p64(a), s0, const, ol.p64(o), C.byref(sc))
        res = ring.decrypt_residues(s, level, o)
        big = pyref.crt_centered([res[m] for m in range(level + 1)], ring.moduli[:level + 1])
        got = np.array([float(x) for x in big]) / sc.value
        want = vo + const * va
        assert np.max(np.abs(got - want)) < 1e-3 * max(1.0, abs(const)), (const, s0, so, float(np.max(np.abs(got - want))))
