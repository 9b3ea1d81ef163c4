"""The C++ host mirror of the reference's Go interface (sfgwas_amd/host/gwas.hpp): CPU logic tests for
GenoFileStream / DiagCacheStream, and a GPU end-to-end test that calls MatMult4Stream, MatMult4StreamPreprocess,
MatMult4StreamCompute and RotateRight the way the Go callers do."""
import ctypes as C
import os
import subprocess
import numpy as np
import pytest

import oracle_lib as ol

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIBDIR = os.path.join(ROOT, "sfgwas_amd", "lib")


def build(name, with_oracle=False):
    """with_oracle: the test program also links the CPU oracle (used as a key generator only)"""
    src = os.path.join(ROOT, "tests", "host", name + ".cpp")
    exe = os.path.join(ROOT, "tests", "host", "_build_" + name)
    hdr = os.path.join(ROOT, "sfgwas_amd", "host", "gwas.hpp")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        odir = os.path.join(ROOT, "oracle", "_build")
        extra = ["-L" + odir, "-loracle", "-Wl,-rpath," + odir] if with_oracle else []
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-o", exe, src, "-L" + LIBDIR, "-lsfgwas_hip", "-Wl,-rpath," + LIBDIR,
                               "-L/opt/rocm/lib", "-Wl,-rpath,/opt/rocm/lib"] + extra)
    return exe


def test_host_logic_and_diagcache_cross_read(tmp_path):
    from sfgwas_amd import capi
    capi.lib()
    exe = build("host_logic_test")
    # a DiagCache file written by the ORACLE must decode identically through the mirror's reader
    L = ol.lib()
    d, n, nmod, vlen = 4, 8, 3, 2
    path = str(tmp_path / "orc_0.bin").encode()
    dc = L.orc_diagcache_create(path, d)
    baby = np.array([1, 1, 0, 0], dtype=np.uint8); giant = np.array([1, 0, 0, 1], dtype=np.uint8)
    L.orc_diagcache_set_tables(dc, baby.ctypes.data_as(C.POINTER(C.c_uint8)), giant.ctypes.data_as(C.POINTER(C.c_uint8)))
    rnd = np.random.default_rng(4)
    digest = (vlen * 1000 + 5 * 100 + nmod) & ((1 << 64) - 1)
    for shift in [1, 5, 12]:
        pv = [rnd.integers(0, 1 << 45, (nmod, n), dtype=np.uint64), None if shift == 5 else rnd.integers(0, 1 << 45, (nmod, n), dtype=np.uint64)]
        arr = (ol.u64p * vlen)(*[ol.p64(p) if p is not None else None for p in pv])
        L.orc_diagcache_write(dc, arr, vlen, 5, 2.0 ** 34, n, nmod, shift)
        digest = (digest * 31 + shift) & ((1 << 64) - 1)
        for p in pv:
            if p is not None:
                for w in p.reshape(-1):
                    digest = (digest * 1099511628211 + int(w)) & ((1 << 64) - 1)
    L.orc_diagcache_close(dc)
    out = subprocess.run([exe, str(tmp_path), str(tmp_path / "orc")], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "OK" in out.stdout
    assert f"DIGEST {digest}" in out.stdout


@pytest.mark.gpu
def test_host_mirror_matmul_end_to_end(tmp_path):
    from sfgwas_amd import capi
    capi.lib()
    exe = build("host_gpu_test")
    ring = ol.Ring(14, ol.Q_PN14, ol.P_PN14)
    keys = ol.RotKeys(ring)
    rnd = np.random.default_rng(21)
    nrow, ncol, s, level, L, square = 45, 33, 2, 5, 5, 0
    geno = rnd.integers(-1, 3, (nrow, ncol)).astype(np.int8)
    geno.tofile(tmp_path / "geno.bin")
    slots, d = 8192, 91
    steps = set()
    for (r, c) in [(nrow, ncol), (ncol, nrow)]:
        for sh in list(range(r)) + list(range(slots - c + 1, slots)):
            if sh % d:
                steps.add(sh % d)
            if sh // d:
                steps.add((sh // d) * d)
    steps.add(slots - 1)                                   # RotateRight(ct, 1) = RotateNew(ct, slots - 1)
    blob = [np.array([len(steps)], dtype=np.uint64)]
    for k in sorted(steps):
        g = ring.galois(k)
        key = capi.random_rotkey(ring.moduli, ring.beta, ring.N, 300 + k)
        keys.add(g, key)
        blob += [np.array([g], dtype=np.uint64), key.reshape(-1)]
    np.concatenate(blob).tofile(tmp_path / "keys.bin")
    np.array([len(ol.Q_PN14), len(ol.P_PN14)] + ol.Q_PN14 + ol.P_PN14, dtype=np.uint64).tofile(tmp_path / "moduli.bin")
    A = np.stack([np.stack([ring.fill_uniform(level, 50 + i)]) for i in range(s)])
    AT = np.stack([np.stack([ring.fill_uniform(level, 70 + i)]) for i in range(s)])
    A.tofile(tmp_path / "A.bin"); AT.tofile(tmp_path / "AT.bin")
    rlk = capi.random_rotkey(ring.moduli, ring.beta, ring.N, 999)
    rlk.tofile(tmp_path / "rlk.bin")
    (tmp_path / "case.txt").write_text(f"{nrow} {ncol} {s} {level} {L} {square}\n")
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr
    want, wsm, wsq = ol.matmult4stream(ring, keys, 2.0 ** 34, A, level, L, geno, compute_sqsum=True)
    got = np.fromfile(tmp_path / "out_stream.bin", dtype=np.uint64).reshape(want.shape)
    assert np.array_equal(got, want)
    sums = np.fromfile(tmp_path / "sums.bin", dtype=np.float64)
    assert np.array_equal(sums[:ncol], wsm) and np.array_equal(sums[ncol:], wsq)
    want_t, _, _ = ol.matmult4stream(ring, keys, 2.0 ** 34, AT, level, L, np.ascontiguousarray(geno.T))
    got_t = np.fromfile(tmp_path / "out_xt.bin", dtype=np.uint64).reshape(want_t.shape)
    assert np.array_equal(got_t, want_t)
    rot = np.fromfile(tmp_path / "rot.bin", dtype=np.uint64).reshape(A[0, 0].shape)
    assert np.array_equal(rot, ol.rotate_right(ring, keys, level, A[0, 0], 1))
    # CMult = MulRelin + one Rescale step at scale 2^68 (basics.go:386-427); CSub (basics.go:580)
    a0, a1 = np.ascontiguousarray(A[0, 0]), np.ascontiguousarray(A[1, 0])
    mr = np.zeros_like(a0)
    ol.lib().orc_mulrelin(ring.h, level, ol.p64(a0), ol.p64(a1), ol.p64(rlk), ol.p64(mr))
    rs = np.zeros((2, level, ring.N), dtype=np.uint64)
    ol.lib().orc_rescale(ring.h, level, ol.p64(mr), ol.p64(rs))
    assert np.array_equal(np.fromfile(tmp_path / "cmult.bin", dtype=np.uint64).reshape(rs.shape), rs)
    df = np.zeros_like(a0)
    ol.lib().orc_ct_addsub(ring.h, level, ol.p64(a0), ol.p64(a1), 1, ol.p64(df))
    assert np.array_equal(np.fromfile(tmp_path / "csub.bin", dtype=np.uint64).reshape(df.shape), df)
    # MaskTrunc(ct, 1000) = encode(1 on the first 1000 slots) x ct, rescaled once
    mvec = np.zeros(ring.slots); mvec[:1000] = 1.0
    mpt = ring.encode_ntt(mvec, 2.0 ** 34, level + 1)
    mp = np.zeros_like(a0)
    ol.lib().orc_mul_plain(ring.h, level, ol.p64(a0), ol.p64(mpt), ol.p64(mp))
    ol.lib().orc_rescale(ring.h, level, ol.p64(mp), ol.p64(rs))
    assert np.array_equal(np.fromfile(tmp_path / "masktrunc.bin", dtype=np.uint64).reshape(rs.shape), rs)
    # CMultConstRescale(X, 1/8192): scaled by q_level, multiplied, rescaled once (scale 2^34 * q5 >= 2^34 * q5 / 2)
    import ctypes as C
    mc = np.zeros_like(a0); sm = C.c_double()
    ol.lib().orc_mul_const(ring.h, level, ol.p64(a0), 1.0 / 8192.0, ol.p64(mc), C.byref(sm))
    assert sm.value == float(ring.moduli[level])
    ol.lib().orc_rescale(ring.h, level, ol.p64(mc), ol.p64(rs))
    assert np.array_equal(np.fromfile(tmp_path / "cmultconst.bin", dtype=np.uint64).reshape(rs.shape), rs)


@pytest.mark.gpu
def test_two_host_threads_share_one_key_set(tmp_path):
    """SURVEY §8b threading: assoc.go:360-408 issues concurrent MatMult4Stream calls; here two host threads, each on its own
    fork of the context (sfg_ctx_fork), run different products at the same time for several rounds — every word vs the oracle"""
    from sfgwas_amd import capi
    capi.lib()
    exe = build("host_concurrent_test")
    ring = ol.Ring(14, ol.Q_PN14, ol.P_PN14)
    keys = ol.RotKeys(ring)
    rnd = np.random.default_rng(31)
    slots, d, level, L = 8192, 91, 5, 5
    cases = [(70, 40, 2, 0), (30, slots + 20, 1, 1)]            # (nrow, ncol, s, square): one block vs two block columns, squared
    steps = set()
    for (r, c, _, _) in cases:
        for c_blk in ([c] if c <= slots else [slots, c - slots]):
            shifts = range(slots) if r + c_blk > slots else list(range(r)) + list(range(slots - c_blk + 1, slots))
            for sh in shifts:
                if sh % d:
                    steps.add(sh % d)
                if sh // d:
                    steps.add((sh // d) * d)
    blob = [np.array([len(steps)], dtype=np.uint64)]
    for k in sorted(steps):
        g = ring.galois(k)
        key = capi.random_rotkey(ring.moduli, ring.beta, ring.N, 300 + k)
        keys.add(g, key)
        blob += [np.array([g], dtype=np.uint64), key.reshape(-1)]
    np.concatenate(blob).tofile(tmp_path / "keys.bin")
    np.array([len(ol.Q_PN14), len(ol.P_PN14)] + ol.Q_PN14 + ol.P_PN14, dtype=np.uint64).tofile(tmp_path / "moduli.bin")
    inputs = []
    for t, (nrow, ncol, s, square) in enumerate(cases):
        geno = rnd.integers(-1, 3, (nrow, ncol)).astype(np.int8)
        geno.tofile(tmp_path / f"geno{t}.bin")
        A = np.stack([np.stack([ring.fill_uniform(level, 500 + 10 * t + i)]) for i in range(s)])
        A.tofile(tmp_path / f"A{t}.bin")
        (tmp_path / f"case{t}.txt").write_text(f"{nrow} {ncol} {s} {level} {L} {square}\n")
        inputs.append((geno, A, square))
    out = subprocess.run([exe, str(tmp_path), str(len(cases)), "3"], capture_output=True, text=True)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr
    for t, (geno, A, square) in enumerate(inputs):
        want, _, _ = ol.matmult4stream(ring, keys, 2.0 ** 34, A, level, L, geno, square=bool(square), enc_prec=1)
        got = np.fromfile(tmp_path / f"out_thr{t}.bin", dtype=np.uint64).reshape(want.shape)
        assert np.array_equal(got, want), f"thread {t}"


# ------------------------------------------------------------------------------------------------ A12 / A13 compositions
def _orc_cmult(ring, level, scale, a, b, rlk, thr=2.0 ** 34):
    """crypto.CMult on two ciphertexts: MulRelinNew + eval.Rescale(ct, params.Scale(), ct); returns (ct, level, scale)"""
    mr = np.zeros_like(a)
    ol.lib().orc_mulrelin(ring.h, level, ol.p64(np.ascontiguousarray(a)), ol.p64(np.ascontiguousarray(b)), ol.p64(rlk), ol.p64(mr))
    return _orc_rescale_loop(ring, mr, level, scale, thr)


def _orc_rescale_loop(ring, ct, level, scale, thr=2.0 ** 34):
    while level != 0 and scale >= thr * float(ring.moduli[level]) / 2:
        rs = np.zeros((2, level, ring.N), dtype=np.uint64)
        ol.lib().orc_rescale(ring.h, level, ol.p64(np.ascontiguousarray(ct)), ol.p64(rs))
        scale /= float(ring.moduli[level]); level -= 1; ct = rs
    return ct, level, scale


def _orc_innersum(ring, keys, level, cts):
    out = np.zeros((2, level + 1, ring.N), dtype=np.uint64)
    cts = np.ascontiguousarray(np.stack(cts))
    assert ol.lib().orc_innersum_all(ring.h, keys.h, level, ol.p64(cts), cts.shape[0], ol.p64(out)) == 0
    return out


def _orc_sub(ring, level, a, b):
    out = np.zeros((2, level + 1, ring.N), dtype=np.uint64)
    ol.lib().orc_ct_addsub(ring.h, level, ol.p64(np.ascontiguousarray(a)), ol.p64(np.ascontiguousarray(b)), 1, ol.p64(out))
    return out


def _drop(ct, level):
    return np.ascontiguousarray(ct[:, :level + 1])


@pytest.mark.gpu
def test_lazy_norm_and_aatb_compositions_stay_on_device_and_match_the_oracle(tmp_path):
    """QXLazyNormStream / QXtLazyNormStream (matmult.go:27-116) and a DCMatMulAAtB column step (matmult.go:121-156) composed from
    device-resident ops in the host mirror, against the same compositions of oracle functions"""
    from sfgwas_amd import capi
    capi.lib()
    exe = build("host_lazynorm_test")
    ring = ol.Ring(14, ol.Q_PN14, ol.P_PN14)
    keys = ol.RotKeys(ring)
    rnd = np.random.default_rng(77)
    nrow, ncol, s, qlevel, slots, d, SC = 45, 33, 2, 7, 8192, 91, 2.0 ** 34
    geno = rnd.integers(-1, 3, (nrow, ncol)).astype(np.int8)
    geno.tofile(tmp_path / "geno.bin")
    steps = set()
    for (r, c) in [(nrow, ncol), (ncol, nrow)]:
        for sh in list(range(r)) + list(range(slots - c + 1, slots)):
            if sh % d:
                steps.add(sh % d)
            if sh // d:
                steps.add((sh // d) * d)
    k = 1
    while k < slots:                                          # InnerSumAll: left rotations by 1, 2, 4, .. (basics.go:236-246)
        steps.add(k); k *= 2
    blob = [np.array([len(steps)], dtype=np.uint64)]
    for k in sorted(steps):
        g = ring.galois(k)
        key = capi.random_rotkey(ring.moduli, ring.beta, ring.N, 300 + k)
        keys.add(g, key)
        blob += [np.array([g], dtype=np.uint64), key.reshape(-1)]
    np.concatenate(blob).tofile(tmp_path / "keys.bin")
    np.array([len(ol.Q_PN14), len(ol.P_PN14)] + ol.Q_PN14 + ol.P_PN14, dtype=np.uint64).tofile(tmp_path / "moduli.bin")
    rlk = capi.random_rotkey(ring.moduli, ring.beta, ring.N, 999)
    rlk.tofile(tmp_path / "rlk.bin")
    Q = np.stack([np.stack([ring.fill_uniform(qlevel, 50 + i)]) for i in range(s)])          # [s][nbr=1]
    Q2 = np.stack([np.stack([ring.fill_uniform(qlevel, 60 + i)]) for i in range(s)])         # [s][m_ct=1]
    XStdInv, XMean = ring.fill_uniform(qlevel, 70), ring.fill_uniform(qlevel - 1, 71)
    XMean2, XStdInv2 = ring.fill_uniform(qlevel, 72), ring.fill_uniform(qlevel, 73)
    for name, a in [("Q", Q), ("Q2", Q2), ("XStdInv", XStdInv), ("XMean", XMean), ("XMean2", XMean2), ("XStdInv2", XStdInv2)]:
        a.tofile(tmp_path / (name + ".bin"))
    (tmp_path / "case.txt").write_text(f"{nrow} {ncol} {s} {qlevel}\n")
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr
    # ---- oracle: QXLazyNormStream
    QS = [_orc_cmult(ring, qlevel, SC * SC, Q[i, 0], XStdInv, rlk) for i in range(s)]
    lvl, sc = QS[0][1], QS[0][2]
    assert lvl == qlevel - 1
    QSm = np.stack([np.stack([q[0]]) for q in QS])
    prod, _, _ = ol.matmult4stream(ring, keys, SC, QSm, lvl, 5, geno, enc_prec=1)
    got1 = np.fromfile(tmp_path / "qx_part1.bin", dtype=np.uint64).reshape(prod.shape)
    assert np.array_equal(got1, prod)
    mask = np.zeros(ring.slots); mask[:((ncol - 1) % slots) + 1] = 1.0
    fin = []
    for i in range(s):
        cm, l2, s2 = _orc_cmult(ring, lvl, sc * SC, QS[i][0], XMean, rlk)
        qsm = _orc_innersum(ring, keys, l2, [cm])
        dct = _r_sub(ring, _Ct(prod[i, 0], 4, sc * SC), _Ct(qsm, l2, s2)).a        # eval.Sub matches the scales first (here: no bootstrap in between, ratio ~2^34)
        mp = np.zeros_like(dct)
        mpt = ring.encode_ntt(mask, SC, 5)
        ol.lib().orc_mul_plain(ring.h, 4, ol.p64(dct), ol.p64(mpt), ol.p64(mp))
        fin.append(_orc_rescale_loop(ring, mp, 4, sc * SC * SC)[0])
    want = np.stack(fin)
    got = np.fromfile(tmp_path / "qx_final.bin", dtype=np.uint64).reshape(want.shape)
    assert np.array_equal(got, want), "QXLazyNormStream local composition"
    # ---- oracle: QXtLazyNormStream
    prod2, _, _ = ol.matmult4stream(ring, keys, SC, Q2, qlevel, 5, np.ascontiguousarray(geno.T), enc_prec=1)
    fin2 = []
    for i in range(s):
        row_sum = _orc_innersum(ring, keys, qlevel, [Q2[i, 0]])
        q1m, l3, s3 = _orc_cmult(ring, qlevel, SC * SC, XMean2, row_sum, rlk)
        dct = _r_sub(ring, _Ct(prod2[i, 0], 4, SC * SC), _Ct(q1m, l3, s3)).a
        fin2.append(_orc_cmult(ring, 4, SC * SC * SC, dct, _drop(XStdInv2, 4), rlk)[0])
    want2 = np.stack(fin2)
    got2 = np.fromfile(tmp_path / "qxt_final.bin", dtype=np.uint64).reshape(want2.shape)
    assert np.array_equal(got2, want2), "QXtLazyNormStream local composition"
    # ---- oracle: DCMatMulAAtB column step, innerFn = CMult(A[c], B[j])
    ctq = []
    for j in range(s):
        cm, l4, s4 = _orc_cmult(ring, qlevel, SC * SC, Q[0, 0], Q[j, 0], rlk)
        ctq.append(_orc_innersum(ring, keys, l4, [cm]))
    got_ctq = np.fromfile(tmp_path / "aatb_ctq.bin", dtype=np.uint64).reshape(np.stack(ctq).shape)
    assert np.array_equal(got_ctq, np.stack(ctq)), "DCMatMulAAtB: cTQloc"
    outs = [_orc_cmult(ring, l4, SC * s4, _drop(Q[0, 0], l4), ctq[j], rlk)[0] for j in range(s)]
    got_out = np.fromfile(tmp_path / "aatb_out.bin", dtype=np.uint64).reshape(np.stack(outs).shape)
    assert np.array_equal(got_out, np.stack(outs)), "DCMatMulAAtB: out[j] += CMult(A[c], cTQ[j])"


# ------------------------------------------------------------------------------------------------ A14 compositions
class OCt:
    """oracle-side ciphertext with lattigo's level / scale bookkeeping (CMult = MulRelin + Rescale(params.Scale()), min-level binary ops)"""
    SC = 2.0 ** 34

    def __init__(self, ring, keys, rlk, data, level, scale):
        self.ring, self.keys, self.rlk, self.data, self.level, self.scale = ring, keys, rlk, np.ascontiguousarray(data), level, scale

    def like(self, data, level, scale):
        return OCt(self.ring, self.keys, self.rlk, data, level, scale)

    def drop(self, level):
        return self if level == self.level else self.like(self.data[:, :level + 1], level, self.scale)

    def rescaled(self):
        d, l, s = _orc_rescale_loop(self.ring, self.data, self.level, self.scale)
        return self.like(d, l, s)

    def cmult(self, o):
        l = min(self.level, o.level); a, b = self.drop(l), o.drop(l)
        mr = np.zeros_like(a.data)
        ol.lib().orc_mulrelin(self.ring.h, l, ol.p64(a.data), ol.p64(b.data), ol.p64(self.rlk), ol.p64(mr))
        return self.like(mr, l, a.scale * b.scale).rescaled()

    def mul_real(self, vals):
        pt = self.ring.encode_ntt(np.asarray(vals, dtype=np.float64), OCt.SC, self.level + 1)
        mp = np.zeros_like(self.data)
        ol.lib().orc_mul_plain(self.ring.h, self.level, ol.p64(self.data), ol.p64(pt), ol.p64(mp))
        return self.like(mp, self.level, self.scale * OCt.SC).rescaled()

    def mask(self, index, keep_rest=False):
        m = np.full(self.ring.slots, 1.0 if keep_rest else 0.0); m[index] = 0.0 if keep_rest else 1.0
        return self.mul_real(m)

    def innersum(self):
        return self.like(_orc_innersum(self.ring, self.keys, self.level, [self.data]), self.level, self.scale)

    def add_const(self, c):
        out = np.zeros_like(self.data)
        ol.lib().orc_add_const(self.ring.h, self.level, ol.p64(self.data), C.c_double(c), C.c_double(self.scale), ol.p64(out))
        return self.like(out, self.level, self.scale)

    def neg(self):
        out = np.zeros_like(self.data); sm = C.c_double(0)
        ol.lib().orc_mul_const(self.ring.h, self.level, ol.p64(self.data), C.c_double(-1.0), ol.p64(out), C.byref(sm))
        assert sm.value == 1.0
        return self.like(out, self.level, self.scale)

    def add(self, o):
        l = min(self.level, o.level); a, b = self.drop(l), o.drop(l)
        out = np.zeros_like(a.data)
        ol.lib().orc_ct_addsub(self.ring.h, l, ol.p64(a.data), ol.p64(b.data), 0, ol.p64(out))
        return self.like(out, l, a.scale)


@pytest.mark.gpu
def test_logistic_path_ciphertext_matrix_helpers_match_the_oracle(tmp_path):
    """CMultMatInnerProd / ...Vector / CMultMatColTimesColToCol / ...RowToCol (matmult.go:1915-2066, used by assoc.go:992-1170)
    as device-resident compositions vs the same compositions of oracle ops, 2 x 2 matrices of single ciphertexts"""
    from sfgwas_amd import capi
    capi.lib()
    exe = build("host_cmat_test")
    ring = ol.Ring(14, ol.Q_PN14, ol.P_PN14)
    keys = ol.RotKeys(ring)
    rows, qlevel, mcols, slots, SC = 2, 7, 5, 8192, 2.0 ** 34
    steps, k = [], 1
    while k < slots:
        steps.append(k); k *= 2
    blob = [np.array([len(steps)], dtype=np.uint64)]
    for k in steps:
        g = ring.galois(k)
        key = capi.random_rotkey(ring.moduli, ring.beta, ring.N, 300 + k)
        keys.add(g, key)
        blob += [np.array([g], dtype=np.uint64), key.reshape(-1)]
    np.concatenate(blob).tofile(tmp_path / "keys.bin")
    np.array([len(ol.Q_PN14), len(ol.P_PN14)] + ol.Q_PN14 + ol.P_PN14, dtype=np.uint64).tofile(tmp_path / "moduli.bin")
    rlk = capi.random_rotkey(ring.moduli, ring.beta, ring.N, 999)
    rlk.tofile(tmp_path / "rlk.bin")
    M = np.stack([np.stack([ring.fill_uniform(qlevel, 150 + i)]) for i in range(rows)])
    Nn = np.stack([np.stack([ring.fill_uniform(qlevel, 160 + i)]) for i in range(rows)])
    M.tofile(tmp_path / "M.bin"); Nn.tofile(tmp_path / "N.bin")
    (tmp_path / "case.txt").write_text(f"{rows} {qlevel} {mcols}\n")
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr
    mk = lambda a: OCt(ring, keys, rlk, a, qlevel, SC)
    Mo, No = [mk(M[i, 0]) for i in range(rows)], [mk(Nn[i, 0]) for i in range(rows)]

    def load(name, like):
        return np.fromfile(tmp_path / name, dtype=np.uint64).reshape(np.stack([x.data for x in like]).shape)

    # CMultMatInnerProd
    res = []
    for r in range(rows):
        acc = None
        for c in range(rows):
            t = Mo[r].cmult(No[c]).innersum().mask(c)
            acc = t if acc is None else t.add(acc)
        res.append(acc)
    assert np.array_equal(load("innerprod.bin", res), np.stack([x.data for x in res])), "CMultMatInnerProd"
    # CMultMatInnerProdVector
    mask_clear = np.zeros(slots); mask_clear[:mcols] = 1.0
    nmask, acc = No[0].mul_real(mask_clear), None
    for kk in range(rows):
        t = Mo[kk].mul_real(mask_clear).cmult(nmask).innersum().mask(kk)
        acc = t if acc is None else t.add(acc)
    assert np.array_equal(load("innerprod_vec.bin", [acc]), acc.data[None]), "CMultMatInnerProdVector"
    # CMultMatColTimesColToCol / RowToCol
    for name, row_form in (("col_col.bin", False), ("col_row.bin", True)):
        res = [None] * rows
        for kk in range(rows):
            for c in range(rows):
                elem = (No[kk].mask(c) if row_form else No[c].mask(kk)).innersum()
                multi = elem.cmult(Mo[kk])
                res[c] = multi if res[c] is None else multi.add(res[c])
        assert np.array_equal(load(name, res), np.stack([x.data for x in res])), name
    # crypto.CInverse = eval.InverseNew(ct, 3) (lattigo v2.1.0 algorithms.go restated, parity unpinned): same composition of oracle ops
    inv = []
    for x in Mo:
        cbar = x.neg().add_const(1.0); res = cbar.add_const(1.0)
        for _ in range(2):
            cbar = cbar.cmult(cbar); res = cbar.add_const(1.0).cmult(res)
        inv.append(res)
    assert inv[0].level == qlevel - 3                       # cbar^2 and the product with res each consume one level per iteration, in parallel
    assert np.array_equal(load("inverse.bin", inv), np.stack([x.data for x in inv])), "CInverse"


@pytest.mark.gpu
def test_collective_bootstrap_local_halves_flatten_and_concat_match_the_oracle(tmp_path):
    """mpc/mhe.go:289-348 (CollectiveBootstrapMat) local halves + crypto.FlattenLevels / ConcatCipherMatrix (basics.go:514-531, 773-790) through the
    host mirror, device resident, vs the oracle's restatement of dckks.RefreshProtocol (parity unpinned: fork source absent)"""
    from sfgwas_amd import capi
    capi.lib()
    exe = build("host_bootstrap_test")
    ring = ol.Ring(14, ol.Q_PN14, ol.P_PN14)
    rows, level, W = 3, 2, 3
    rnd = np.random.default_rng(21)
    np.array([len(ol.Q_PN14), len(ol.P_PN14)] + ol.Q_PN14 + ol.P_PN14, dtype=np.uint64).tofile(tmp_path / "moduli.bin")
    sk = ol.secret_ntt(ring, ring.gen_secret(6)); sk.tofile(tmp_path / "sk.bin")
    first_hi = ring.fill_uniform(level + 1, 301)
    rest = np.stack([ring.fill_uniform(level, 310 + i) for i in range(rows - 1)])
    first_hi.tofile(tmp_path / "cm_first_hi.bin"); rest.tofile(tmp_path / "cm_rest.bin")
    cm = np.concatenate([first_hi[None, :, :level + 1, :], rest])                 # DropLevel keeps the first level+1 rows of both polynomials
    Ql = 1
    for q in ring.moduli[:level + 1]:
        Ql *= q
    bound = Ql // 4
    limbs = np.zeros((rows, ring.N, W), dtype=np.uint64)
    for i in range(rows):
        vals = []
        for _ in range(ring.N):
            m = int.from_bytes(rnd.bytes(40), "little") % bound
            vals.append(m - bound if m >= bound >> 1 else m)
        limbs[i] = ol.bigints_to_limbs(vals, W)
    limbs.tofile(tmp_path / "mask.bin")
    crs = np.stack([np.stack([rnd.integers(0, ring.moduli[j], ring.N, dtype=np.uint64) for j in range(ring.nq)]) for _ in range(rows)])
    crs.tofile(tmp_path / "crs.bin")
    e0 = rnd.integers(-19, 20, (rows, ring.N)).astype(np.int32); e1 = rnd.integers(-19, 20, (rows, ring.N)).astype(np.int32)
    np.concatenate([e0.reshape(-1), e1.reshape(-1)]).tofile(tmp_path / "e.bin")
    h0agg = np.stack([np.stack([rnd.integers(0, ring.moduli[j], ring.N, dtype=np.uint64) for j in range(level + 1)]) for _ in range(rows)])
    h1agg = np.stack([np.stack([rnd.integers(0, ring.moduli[j], ring.N, dtype=np.uint64) for j in range(ring.nq)]) for _ in range(rows)])
    h0agg.tofile(tmp_path / "h0agg.bin"); h1agg.tofile(tmp_path / "h1agg.bin")
    ct_scale = 2.0 ** 68 / ring.moduli[5] * 2.0 ** 34            # CMult(Q, XStdInv) rescaled once, then the product's Delta (matmult.go:36-44): not a power of two
    (tmp_path / "case.txt").write_text(f"{rows} {level} {W} {ct_scale!r}\n")
    # eval.MultByConstAndAdd cases (pca.go:264: integer constant; qrfact.go:195,280: -2/N): (constant, level0, scale0, levelOut, scaleOut)
    SC = 2.0 ** 34
    mb = [(-3.0, 4, SC, 4, SC), (-2.0 / 3000.0, 4, SC * 1.0001, 4, SC), (5.0, 3, SC, 5, SC * 64.0), (-7.0, 4, SC * 1000.0, 4, SC), (0.375, 2, SC, 2, SC * 2.0 ** 40)]
    (tmp_path / "mbca_cases.txt").write_text(f"{len(mb)}\n" + "".join(f"{c!r} {l0} {s0!r} {lo} {so!r}\n" for c, l0, s0, lo, so in mb))
    mb_in, mb_out = [], []
    for k, (c, l0, s0, lo, so) in enumerate(mb):
        a = np.stack([ring.fill_uniform(l0, 700 + 2 * k + j) for j in range(2)]); o = np.stack([ring.fill_uniform(lo, 800 + 2 * k + j) for j in range(2)])
        a.tofile(tmp_path / f"mbca_in_{k}.bin"); o.tofile(tmp_path / f"mbca_out_{k}.bin"); mb_in.append(a); mb_out.append(o)
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr
    ld = lambda name, shape: np.fromfile(tmp_path / name, dtype=np.uint64).reshape(shape)
    got_scales = [ln.split() for ln in (tmp_path / "mbca_scales.txt").read_text().splitlines()]
    for k, (c, l0, s0, lo, so) in enumerate(mb):
        lev = min(l0, lo)
        res = ld(f"mbca_res_{k}.bin", (2, 2, lev + 1, ring.N))
        for j in range(2):
            a = np.ascontiguousarray(mb_in[k][j][:, :lev + 1]); o = np.ascontiguousarray(mb_out[k][j][:, :lev + 1])     # DropLevel keeps the first lev+1 rows
            sc = C.c_double(so)
            ol.lib().orc_mul_const_and_add(ring.h, lev, ol.p64(a), s0, c, ol.p64(o), C.byref(sc))
            assert np.array_equal(res[j], o), f"MultByConstAndAdd case {k} ciphertext {j}"
        assert int(got_scales[k][0]) == lev and float(got_scales[k][1]) == sc.value, f"MultByConstAndAdd case {k}: level / scale bookkeeping"
    assert np.array_equal(ld("flat.bin", cm.shape), cm), "FlattenLevels"
    cc = ld("concat.bin", (rows, 2) + cm.shape[1:])
    assert np.array_equal(cc[:, 0], cm) and np.array_equal(cc[:, 1], cm), "ConcatCipherMatrix"
    h0, h1 = ld("h0.bin", (rows, level + 1, ring.N)), ld("h1.bin", (rows, ring.nq, ring.N))
    got = ld("out.bin", (rows, 2, ring.nq, ring.N))
    for i in range(rows):
        w0, w1 = ol.refresh_gen_shares_scaled(ring, level, cm[i], ct_scale, 2.0 ** 34, sk, crs[i], limbs[i], e0[i], e1[i])
        assert np.array_equal(h0[i], w0) and np.array_equal(h1[i], w1), f"GenShares of ciphertext {i}"
        assert np.array_equal(got[i], ol.refresh_finish_scaled(ring, level, cm[i], ct_scale, 2.0 ** 34, h0agg[i], h1agg[i], crs[i])), f"Decrypt/Recode/Recrypt of ciphertext {i}"


# ---------------------------------------------------------------- one power iteration's local segments, real keys, bootstraps at the target scale
class _Ct:
    """oracle-side ciphertext with lattigo's bookkeeping: residues [2][level+1][N], level, scale"""
    def __init__(self, a, level, scale):
        self.a, self.level, self.scale = np.ascontiguousarray(a), level, scale


def _r_drop(x, level):
    return x if x.level == level else _Ct(_drop(x.a, level), level, x.scale)


def _r_cmult(ring, x, y, rlk, SC):
    """crypto.CMult: MulRelinNew at the lower level + eval.Rescale(ct, Params.Scale(), ct)"""
    l = min(x.level, y.level)
    ct, lvl, sc = _orc_cmult(ring, l, x.scale * y.scale, _r_drop(x, l).a, _r_drop(y, l).a, rlk, SC)
    return _Ct(ct, lvl, sc)


def _r_mul_int(ring, x, k, new_scale):
    out = np.zeros_like(x.a); sm = C.c_double(0)
    ol.lib().orc_mul_const(ring.h, x.level, ol.p64(x.a), float(k), ol.p64(out), C.byref(sm))
    assert sm.value == 1.0                                            # an integer constant does not change the scale
    return _Ct(out, x.level, new_scale)


def _r_sub(ring, x, y):
    """eval.Sub with lattigo's scale matching (the operand with the smaller scale times floor(ratio) when > 1; result at the larger scale)"""
    l = min(x.level, y.level)
    x, y = _r_drop(x, l), _r_drop(y, l)
    out_scale = x.scale
    if x.scale > y.scale and np.floor(x.scale / y.scale) > 1:
        y = _r_mul_int(ring, y, np.floor(x.scale / y.scale), x.scale)
    elif y.scale > x.scale and np.floor(y.scale / x.scale) > 1:
        x = _r_mul_int(ring, x, np.floor(y.scale / x.scale), y.scale); out_scale = y.scale
    return _Ct(_orc_sub(ring, l, x.a, y.a), l, out_scale)


def _r_innersum(ring, keys, xs):
    return _Ct(_orc_innersum(ring, keys, xs[0].level, [x.a for x in xs]), xs[0].level, xs[0].scale)


def _r_masktrunc(ring, x, nkeep, SC):
    if nkeep == ring.slots:
        return x
    mask = np.zeros(ring.slots); mask[:nkeep] = 1.0
    mp = np.zeros_like(x.a)
    mpt = ring.encode_ntt(mask, SC, x.level + 1)
    ol.lib().orc_mul_plain(ring.h, x.level, ol.p64(x.a), ol.p64(mpt), ol.p64(mp))
    ct, lvl, sc = _orc_rescale_loop(ring, mp, x.level, x.scale * SC, SC)
    return _Ct(ct, lvl, sc)


def _r_bootstrap(ring, x, sk, rnd_set, k, SC):
    """one party: its own shares are the aggregate (mhe.go:313-331 at the target scale Params.Scale())"""
    masks, crs, e0, e1 = rnd_set
    h0, h1 = ol.refresh_gen_shares_scaled(ring, x.level, x.a, x.scale, SC, sk, crs[k], masks[k], e0[k], e1[k])
    return _Ct(ol.refresh_finish_scaled(ring, x.level, x.a, x.scale, SC, h0, h1, crs[k]), ring.nq - 1, SC)


@pytest.mark.gpu
def test_power_iteration_local_segments_with_bootstraps_at_two_block_rows_and_columns(tmp_path):
    """pca.go:339-353 local work at n_ind = 8192 + 45, m_snp = 8192 + 33 (nbr = m_ct = 2, ragged second blocks), kp = 2, REAL keys from a toy secret:
    QXtLazyNormStream part 1 -> bootstrap (level 4 -> 9, scale A.scale * Delta -> Delta) -> part 2 -> bootstrap -> QXLazyNormStream part 1 -> bootstrap
    -> part 2 with MaskTrunc on the ragged column only: the result carries per-ciphertext levels and scales (matmult.go:60-70).
    Inputs are fresh (level 9), so the reference's own corner applies: CMult at level 9 does not rescale (q_9 > 2^35) and eval.Sub has to match a
    scale-2^68 operand with a scale-2^34 one.
    (a) every dumped word, level and scale equals the oracle's replay of the same composition; (b) the final cells DECRYPT, at their own scales, to
    (QS X^T - (QS m) 1^T) with zeros in the tail slots - the decode-level check of the scale bookkeeping."""
    import pyref
    from sfgwas_amd import capi
    capi.lib()
    ol.build_oracle()
    exe = build("host_poweriter_test", with_oracle=True)
    ring = ol.Ring(14, ol.Q_PN14, ol.P_PN14)
    slots, d, SC, top = 8192, 91, 2.0 ** 34, 9
    n_ind, m_snp, s, W, seed = slots + 45, slots + 33, 2, 4, 31337
    nbr, mct = 2, 2
    rnd = np.random.default_rng(2024)
    geno = rnd.integers(-1, 3, (n_ind, m_snp)).astype(np.int8)
    geno.tofile(tmp_path / "geno.bin")
    G = np.where(geno < 0, 0, geno).astype(np.float64)
    sec = ring.gen_secret(seed)
    sk = ol.secret_ntt(ring, sec)
    keys = ol.RotKeys(ring)
    steps = sorted(set(range(1, d)) | {g * d for g in range(1, d) if g * d < slots} | {1 << k for k in range(13)})
    for k in steps:
        g = ring.galois(k)
        keys.add(g, ring.gen_rotkey(sec, g, 5000 + k))
    rlk = np.zeros(ring.key_words(), dtype=np.uint64)
    ol.lib().orc_gen_rlk(ring.h, ol.pi8(sec), 4999, ol.p64(rlk))
    np.array([len(ol.Q_PN14), len(ol.P_PN14)] + ol.Q_PN14 + ol.P_PN14, dtype=np.uint64).tofile(tmp_path / "moduli.bin")
    (tmp_path / "case.txt").write_text(f"{n_ind} {m_snp} {s} {W} {seed}\n")
    Qp = rnd.normal(size=(s, n_ind)) / 64.0
    mean = G.mean(0)
    sinv = 1.0 / np.maximum(G.std(0), 0.25)

    def enc_vec(v, n_ct, seed0):
        pad = np.zeros(n_ct * slots); pad[:len(v)] = v
        return np.stack([ring.encrypt(sec, top, ring.encode_coeffs(pad[k * slots:(k + 1) * slots], SC), seed0 + k) for k in range(n_ct)])
    Q = np.stack([enc_vec(Qp[i], nbr, 100 + 10 * i) for i in range(s)])
    XMean, XStdInv = enc_vec(mean, mct, 300), enc_vec(sinv, mct, 400)
    Q.tofile(tmp_path / "Q.bin"); XMean.tofile(tmp_path / "XMean.bin"); XStdInv.tofile(tmp_path / "XStdInv.bin")

    def randomness(tag, nct, level):
        Ql = 1
        for q in ring.moduli[:level + 1]:
            Ql *= q
        bound = Ql // 4
        masks = np.zeros((nct, ring.N, W), dtype=np.uint64)
        for c in range(nct):
            vals = []
            for _ in range(ring.N):
                m = int.from_bytes(rnd.bytes(40), "little") % bound
                vals.append(m - bound if m >= bound >> 1 else m)
            masks[c] = ol.bigints_to_limbs(vals, W)
        crs = np.stack([np.stack([rnd.integers(0, ring.moduli[j], ring.N, dtype=np.uint64) for j in range(ring.nq)]) for _ in range(nct)])
        e0 = rnd.integers(-19, 20, (nct, ring.N)).astype(np.int32); e1 = rnd.integers(-19, 20, (nct, ring.N)).astype(np.int32)
        masks.tofile(tmp_path / f"{tag}_mask.bin"); crs.tofile(tmp_path / f"{tag}_crs.bin")
        np.concatenate([e0.reshape(-1), e1.reshape(-1)]).tofile(tmp_path / f"{tag}_e.bin")
        return masks, crs, e0, e1
    # ---------------- oracle replay first (it fixes the level the second bootstrap's masks must be drawn for), A
    cQ = [[_Ct(Q[i, b], top, SC) for b in range(nbr)] for i in range(s)]
    cMean, cSinv = [_Ct(x, top, SC) for x in XMean], [_Ct(x, top, SC) for x in XStdInv]
    prodA, _, _ = ol.matmult4stream(ring, keys, SC, Q, top, 5, geno, enc_prec=1)
    rA = randomness("bootA", s * mct, 4)
    bootA = [[_r_bootstrap(ring, _Ct(prodA[i, j], 4, SC * SC), sk, rA, i * mct + j, SC) for j in range(mct)] for i in range(s)]
    outA = []
    for i in range(s):
        row_sum = _r_innersum(ring, keys, cQ[i])
        q1m = [_r_cmult(ring, cMean[j], row_sum, rlk, SC) for j in range(mct)]
        outA.append([_r_cmult(ring, _r_sub(ring, bootA[i][j], q1m[j]), cSinv[j], rlk, SC) for j in range(mct)])
    lA, scA = outA[0][0].level, outA[0][0].scale
    assert q1m[0].level == top and q1m[0].scale == SC * SC, "the corner this test is about: CMult at level 9 must not rescale"
    rM = randomness("bootM", s * mct, lA)
    Q1 = [[_r_bootstrap(ring, outA[i][j], sk, rM, i * mct + j, SC) for j in range(mct)] for i in range(s)]
    # B
    QS = [[_r_cmult(ring, Q1[i][j], cSinv[j], rlk, SC) for j in range(mct)] for i in range(s)]
    lQS, scQS = QS[0][0].level, QS[0][0].scale
    prodB, _, _ = ol.matmult4stream(ring, keys, SC, np.stack([np.stack([c.a for c in QS[i]]) for i in range(s)]), lQS, 5, np.ascontiguousarray(geno.T), enc_prec=1)
    rB = randomness("bootB", s * nbr, 4)
    bootB = [[_r_bootstrap(ring, _Ct(prodB[i, j], 4, scQS * SC), sk, rB, i * nbr + j, SC) for j in range(nbr)] for i in range(s)]
    fin = []
    for i in range(s):
        qsm = _r_innersum(ring, keys, [_r_cmult(ring, QS[i][j], cMean[j], rlk, SC) for j in range(mct)])
        fin.append([_r_masktrunc(ring, _r_sub(ring, bootB[i][j], qsm), slots if j + 1 < nbr else ((n_ind - 1) % slots) + 1, SC) for j in range(nbr)])
    # ---------------- the device-resident chain
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-3000:]
    meta = {ln.split()[0]: ln.split()[1:] for ln in (tmp_path / "meta.txt").read_text().splitlines()}

    def check(name, cells):
        r, c, lvl, sc = int(meta[name][0]), int(meta[name][1]), int(meta[name][2]), float(meta[name][3])
        got = np.fromfile(tmp_path / (name + ".bin"), dtype=np.uint64).reshape(r, c, 2, lvl + 1, ring.N)
        for i in range(r):
            for j in range(c):
                w = cells[i][j]
                assert (lvl, sc) == (w.level, w.scale), f"{name}[{i}][{j}]: level / scale ({lvl}, {sc!r}) vs the replay's ({w.level}, {w.scale!r})"
                assert np.array_equal(got[i, j], w.a), f"{name}[{i}][{j}]: words"
        return got
    check("a_prod", [[_Ct(prodA[i, j], 4, SC * SC) for j in range(mct)] for i in range(s)])
    check("a_boot", bootA)
    check("a_out", outA)
    check("q1", Q1)
    check("b_prod", [[_Ct(prodB[i, j], 4, scQS * SC) for j in range(nbr)] for i in range(s)])
    check("b_boot", bootB)
    # plaintext of the whole chain
    R1 = (Qp @ G - np.outer(Qp.sum(1), mean)) * sinv
    QSp = R1 * sinv
    R2 = QSp @ G.T - np.outer(QSp @ mean, np.ones(n_ind))
    for i in range(s):
        for j in range(nbr):
            name = f"b_out_{i}_{j}"
            got = check(name, [[fin[i][j]]])
            lvl, sc = fin[i][j].level, fin[i][j].scale
            res = ring.decrypt_residues(sec, lvl, got[0, 0])                       # decode-level check at the ciphertext's OWN scale
            big = pyref.crt_centered([res[m] for m in range(4)], ring.moduli[:4])
            dec = pyref.decode(np.array([float(x) for x in big]) / sc, ring.N).real
            ref = np.zeros(slots); seg = R2[i, j * slots:(j + 1) * slots]; ref[:len(seg)] = seg
            assert np.max(np.abs(dec - ref)) < 2e-3 * max(1.0, np.max(np.abs(R2))), f"{name}: decrypted values off by {np.max(np.abs(dec - ref))}"
    assert fin[0][0].level == fin[0][1].level + 1 and fin[0][0].scale != fin[0][1].scale, "the full column stays one level above the masked tail, at its own scale"


# ---------------------------------------------------------------- f-2: NetDQRenc local segments (forward + backward column), two ciphertexts per column
def _r_add(ring, x, y):
    """eval.Add with lattigo's scale matching"""
    l = min(x.level, y.level)
    x, y = _r_drop(x, l), _r_drop(y, l)
    out_scale = x.scale
    if x.scale > y.scale and np.floor(x.scale / y.scale) > 1:
        y = _r_mul_int(ring, y, np.floor(x.scale / y.scale), x.scale)
    elif y.scale > x.scale and np.floor(y.scale / x.scale) > 1:
        x = _r_mul_int(ring, x, np.floor(y.scale / x.scale), y.scale); out_scale = y.scale
    out = np.zeros_like(x.a)
    ol.lib().orc_ct_addsub(ring.h, l, ol.p64(x.a), ol.p64(y.a), 0, ol.p64(out))
    return _Ct(out, l, out_scale)


def _r_innersum_cells(ring, keys, xs):
    """crypto.InnerSumAll: vecsum = X[0]; eval.Add(X[i], vecsum, vecsum); then the rotate-and-add ladder"""
    vs = xs[0]
    for x in xs[1:]:
        vs = _r_add(ring, vs, x)
    return _Ct(_orc_innersum(ring, keys, vs.level, [vs.a]), vs.level, vs.scale)


def _r_mul_const(ring, x, constant):
    """eval.MultByConstNew: a fractional constant is scaled by q_level"""
    out = np.zeros_like(x.a); sm = C.c_double(0)
    ol.lib().orc_mul_const(ring.h, x.level, ol.p64(x.a), float(constant), ol.p64(out), C.byref(sm))
    return _Ct(out, x.level, x.scale * sm.value)


def _r_mask(ring, x, index, keep_rest, SC):
    m = np.full(ring.slots, 1.0 if keep_rest else 0.0); m[index] = 0.0 if keep_rest else 1.0
    mp = np.zeros_like(x.a)
    mpt = ring.encode_ntt(m, SC, x.level + 1)
    ol.lib().orc_mul_plain(ring.h, x.level, ol.p64(x.a), ol.p64(mpt), ol.p64(mp))
    ct, lvl, sc = _orc_rescale_loop(ring, mp, x.level, x.scale * SC, SC)
    return _Ct(ct, lvl, sc)


def _r_rebalance(ring, keys, x):
    return _r_mul_const(ring, _r_innersum_cells(ring, keys, [x]), 1.0 / ring.slots)


def _r_mbca(ring, ct0, constant, out):
    """eval.MultByConstAndAdd(ct0, constant, ctOut): both at the lower level, scales matched by the restated rule (orc_mul_const_and_add)"""
    l = min(ct0.level, out.level)
    a, o = _r_drop(ct0, l).a.copy(), _r_drop(out, l).a.copy()
    sc = C.c_double(out.scale)
    ol.lib().orc_mul_const_and_add(ring.h, l, ol.p64(a), ct0.scale, float(constant), ol.p64(o), C.byref(sc))
    return _Ct(o, l, sc.value)


@pytest.mark.gpu
def test_netdqrenc_forward_and_backward_column_segments_match_the_oracle(tmp_path):
    """qrfact.go:75-216 (forward column: SqSum, Householder vector from alphaScaled / zNewSqrtInv, DCMatMulAAtB halves, MultByConstAndAdd(-2/N),
    bootstrap at the target scale, pivot-row Mask, FlattenLevels) and :236-285 (backward column on a Q slice) as device-resident sequences, columns of
    two ciphertexts, pivot in the SECOND ciphertext - so the Householder vector has mixed levels.  Every dumped word, level and scale vs the oracle replay."""
    from sfgwas_amd import capi
    capi.lib()
    exe = build("host_qr_test")
    ring = ol.Ring(14, ol.Q_PN14, ol.P_PN14)
    keys = ol.RotKeys(ring)
    rnd = np.random.default_rng(99)
    ncols, nct, ctid, slotid, W, level, totN, SC, slots = 3, 2, 1, 5, 4, 9, 20000.0, 2.0 ** 34, 8192
    blob = [np.array([13], dtype=np.uint64)]
    for k in [1 << t for t in range(13)]:                       # InnerSumAll: left rotations by 2^k (basics.go:236-246)
        g = ring.galois(k)
        key = capi.random_rotkey(ring.moduli, ring.beta, ring.N, 700 + k)
        keys.add(g, key)
        blob += [np.array([g], dtype=np.uint64), key.reshape(-1)]
    np.concatenate(blob).tofile(tmp_path / "keys.bin")
    np.array([len(ol.Q_PN14), len(ol.P_PN14)] + ol.Q_PN14 + ol.P_PN14, dtype=np.uint64).tofile(tmp_path / "moduli.bin")
    rlk = capi.random_rotkey(ring.moduli, ring.beta, ring.N, 998); rlk.tofile(tmp_path / "rlk.bin")
    sk = ol.secret_ntt(ring, ring.gen_secret(8)); sk.tofile(tmp_path / "sk.bin")
    A = np.stack([np.stack([ring.fill_uniform(level, 1000 + 10 * c + ci) for ci in range(nct)]) for c in range(ncols)])
    Q = np.stack([np.stack([ring.fill_uniform(level, 2000 + 10 * c + ci) for ci in range(nct)]) for c in range(ncols)])
    alpha, zinv = ring.fill_uniform(level, 3001), ring.fill_uniform(level, 3002)
    A.tofile(tmp_path / "A.bin"); Q.tofile(tmp_path / "Q.bin"); alpha.tofile(tmp_path / "alpha.bin"); zinv.tofile(tmp_path / "zinv.bin")
    (tmp_path / "case.txt").write_text(f"{ncols} {nct} {ctid} {slotid} {W} {level} {totN!r}\n")
    # ---- oracle replay up to the bootstrap (its level fixes the mask bound)
    cA = [[_Ct(A[c, ci], level, SC) for ci in range(nct)] for c in range(ncols)]
    cQ = [[_Ct(Q[c, ci], level, SC) for ci in range(nct)] for c in range(ncols)]
    want = {}
    want["f1_zloc"] = _r_innersum_cells(ring, keys, [_r_cmult(ring, x, x, rlk, SC) for x in cA[0]])
    al = _r_mask(ring, _r_rebalance(ring, keys, _Ct(alpha, level, SC)), slotid, False, SC)
    zi = _r_rebalance(ring, keys, _Ct(zinv, level, SC))
    uvec = [_r_cmult(ring, x, zi, rlk, SC) for x in cA[0]]
    uvec[ctid] = _r_add(ring, uvec[ctid], _r_mask(ring, al, slotid, False, SC))
    for ci in range(nct):
        want[f"f2_uvec_{ci}"] = uvec[ci]
    assert uvec[0].level != uvec[1].level, "the pivot ciphertext of the Householder vector sits below the other one: the mixed-level case"

    def inner(v, B, backward):
        out = []
        for j in range(len(B)):
            if backward and j == 0:
                out.append(_r_innersum_cells(ring, keys, [_r_mask(ring, v[ctid], slotid, False, SC)]))
            else:
                out.append(_r_innersum_cells(ring, keys, [_r_cmult(ring, v[ci], B[j][ci], rlk, SC) for ci in range(nct)]))
        return out

    def update(v, cTQ, M, consts):
        for j in range(len(cTQ)):
            for ci in range(nct):
                M[j][ci] = _r_mbca(ring, _r_cmult(ring, v[ci], cTQ[j], rlk, SC), consts[j], M[j][ci])
    ctq = inner(uvec, cA, False)
    for j in range(ncols):
        want[f"f3_ctq_{j}"] = ctq[j]
    update(uvec, ctq, cA, [-2.0 / totN] * ncols)
    for c in range(ncols):
        for ci in range(nct):
            want[f"f3_A_{c}_{ci}"] = cA[c][ci]
    lvl = min(x.level for col in cA for x in col)
    Ql = 1
    for q in ring.moduli[:lvl + 1]:
        Ql *= q
    bound = Ql // 4
    nb = ncols * nct
    masks = np.zeros((nb, ring.N, W), dtype=np.uint64)
    for k in range(nb):
        vals = []
        for _ in range(ring.N):
            m = int.from_bytes(rnd.bytes(40), "little") % bound
            vals.append(m - bound if m >= bound >> 1 else m)
        masks[k] = ol.bigints_to_limbs(vals, W)
    crs = np.stack([np.stack([rnd.integers(0, ring.moduli[j], ring.N, dtype=np.uint64) for j in range(ring.nq)]) for _ in range(nb)])
    e0 = rnd.integers(-19, 20, (nb, ring.N)).astype(np.int32); e1 = rnd.integers(-19, 20, (nb, ring.N)).astype(np.int32)
    masks.tofile(tmp_path / "bootF_mask.bin"); crs.tofile(tmp_path / "bootF_crs.bin")
    np.concatenate([e0.reshape(-1), e1.reshape(-1)]).tofile(tmp_path / "bootF_e.bin")
    Ab = [[_r_bootstrap(ring, _r_drop(cA[c][ci], lvl), sk, (masks, crs, e0, e1), c * nct + ci, SC) for ci in range(nct)] for c in range(ncols)]
    A4 = [list(col) for col in Ab[1:]]
    for col in A4:
        col[ctid] = _r_mask(ring, col[ctid], slotid, True, SC)
    l4 = min(x.level for col in A4 for x in col)
    for c, col in enumerate(A4):
        for ci in range(nct):
            want[f"f4_A_{c}_{ci}"] = _r_drop(col[ci], l4)
    ctqb = inner(uvec, cQ, True)
    for j in range(ncols):
        want[f"b_ctq_{j}"] = ctqb[j]
    update(uvec, ctqb, cQ, [-2.0 / np.sqrt(totN)] + [-2.0 / totN] * (ncols - 1))
    for c in range(ncols):
        for ci in range(nct):
            want[f"b_Q_{c}_{ci}"] = cQ[c][ci]
    # ---- the device-resident sequence
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-3000:]
    meta = {ln.split()[0]: ln.split()[1:] for ln in (tmp_path / "meta.txt").read_text().splitlines()}
    assert set(meta) == set(want)
    for name, w in want.items():
        lvl_, sc_ = int(meta[name][0]), float(meta[name][1])
        assert (lvl_, sc_) == (w.level, w.scale), f"{name}: level / scale ({lvl_}, {sc_!r}) vs the replay's ({w.level}, {w.scale!r})"
        got = np.fromfile(tmp_path / (name + ".bin"), dtype=np.uint64).reshape(2, lvl_ + 1, ring.N)
        assert np.array_equal(got, w.a), f"{name}: words"


# ---------------------------------------------------------------- f-5: CSigmoidApprox = change of variable + eval.EvaluateCheby (mhe.go:634-667), parity unpinned
def _r_add_const(ring, x, c):
    out = np.zeros_like(x.a)
    ol.lib().orc_add_const(ring.h, x.level, ol.p64(x.a), C.c_double(c), C.c_double(x.scale), ol.p64(out))
    return _Ct(out, x.level, x.scale)


def _r_rescale_once(ring, x, SC):
    ct, lvl, sc = _orc_rescale_loop(ring, x.a, x.level, x.scale, SC)
    assert lvl == x.level - 1
    return _Ct(ct, lvl, sc)


def _r_mulrelin(ring, x, y, rlk):
    l = min(x.level, y.level)
    a, b = _r_drop(x, l).a, _r_drop(y, l).a
    mr = np.zeros_like(a)
    ol.lib().orc_mulrelin(ring.h, l, ol.p64(a), ol.p64(b), ol.p64(rlk), ol.p64(mr))
    return _Ct(mr, l, x.scale * y.scale)


class _Poly:
    def __init__(self, c, max_deg, lead):
        self.c, self.max_deg, self.lead = list(c), max_deg, lead

    @property
    def deg(self):
        return len(self.c) - 1


def _cheby_split(p, split):
    r = _Poly(p.c[:split], split - 1 if p.max_deg == p.deg else p.max_deg - (p.deg - split + 1), False)
    q = _Poly([0.0] * (p.deg - split + 1), p.max_deg, p.lead)
    q.c[0] = p.c[split]
    for j, i in enumerate(range(split + 1, p.deg + 1), start=1):
        q.c[i - split] = 2 * p.c[i]
        r.c[split - j] -= p.c[i]
    return q, r


class _ChebyReplay:
    """lattigo v2.1 / v2.2 EvaluateCheby restated a second time, on oracle ciphertexts (the device mirror is sfgwas_amd/host/gwas.hpp detail::ChebyEval)"""
    def __init__(self, ring, rlk, SC, op):
        self.ring, self.rlk, self.SC, self.C = ring, rlk, SC, {1: op}
        self.n_mul = 0

    def power(self, n):
        if n in self.C:
            return
        a, b = (n + 1) // 2, n // 2
        c = a - b
        self.power(a); self.power(b)
        if c:
            self.power(c)
        t = _r_cmult(self.ring, self.C[a], self.C[b], self.rlk, self.SC); self.n_mul += 1
        t = _r_add(self.ring, t, t)
        self.C[n] = _r_add_const(self.ring, t, -1.0) if c == 0 else _r_sub(self.ring, t, self.C[c])

    @staticmethod
    def resplit(p, log_split):
        return p.lead and log_split > 1 and p.max_deg % (1 << (log_split + 1)) > (1 << (log_split - 1))

    @staticmethod
    def next_power(p, log_split):
        np_ = 1 << log_split
        while np_ < (p.deg >> 1) + 1:
            np_ <<= 1
        return np_

    def out_level(self, p, log_split, log_degree):
        """(level of the result, level of the MulRelin with the power-basis element)"""
        if p.deg < (1 << log_split):
            if self.resplit(p, log_split):
                ld = p.deg.bit_length()
                return self.out_level(p, ld >> 1, ld)
            return self.C[p.deg or 1].level - 1, None
        np_ = self.next_power(p, log_split)
        q, r = _cheby_split(p, np_)
        lq, lr = self.out_level(q, log_split, log_degree)[0], self.out_level(r, log_split, log_degree)[0]
        if lq > lr:
            lq = lr + 1
        lm = min(lq, self.C[np_].level)
        return (min(lm - 1, lr) if lm > lr else min(lm, lr) - 1), lm

    def leaf(self, target, p):
        top = self.C[p.deg or 1]
        qi = float(self.ring.moduli[top.level])
        res = _Ct(np.zeros_like(top.a), top.level, target * qi)
        if abs(p.c[0]) > 1e-14:
            res = _r_add_const(self.ring, res, p.c[0])
        for key in range(p.deg, 0, -1):
            if not abs(p.c[key]) > 1e-14:
                continue
            T = self.C[key]
            c_real = int(p.c[key] * (target * qi / T.scale))                  # toward zero, as Go's int64()
            term = _r_mul_int(self.ring, _r_drop(T, top.level), c_real, res.scale)
            out = np.zeros_like(res.a)
            ol.lib().orc_ct_addsub(self.ring.h, top.level, ol.p64(res.a), ol.p64(term.a), 0, ol.p64(out))
            res = _Ct(out, top.level, res.scale)
        return _r_rescale_once(self.ring, res, self.SC)

    def recurse(self, target, log_split, log_degree, p):
        if p.deg < (1 << log_split):
            if self.resplit(p, log_split):
                ld = p.deg.bit_length()
                return self.recurse(target, ld >> 1, ld, p)
            return self.leaf(target, p)
        np_ = self.next_power(p, log_split)
        q, r = _cheby_split(p, np_)
        lm = self.out_level(p, log_split, log_degree)[1]
        T = self.C[np_]
        res = self.recurse(target * float(self.ring.moduli[lm]) / T.scale, log_split, log_degree, q)
        tmp = self.recurse(target, log_split, log_degree, r)
        if res.level > tmp.level:
            res = _r_drop(res, tmp.level + 1)
        res = _r_mulrelin(self.ring, res, T, self.rlk); self.n_mul += 1
        assert res.level == lm
        if res.level > tmp.level:
            return _r_add(self.ring, _r_rescale_once(self.ring, res, self.SC), tmp)
        return _r_rescale_once(self.ring, _r_add(self.ring, res, tmp), self.SC)

    def evaluate(self, coeffs, target):
        p = _Poly(coeffs, len(coeffs) - 1, True)
        log_degree = p.deg.bit_length(); log_split = log_degree >> 1
        for i in range(2, 1 << log_split):
            self.power(i)
        for i in range(log_split, log_degree):
            self.power(1 << i)
        return self.recurse(target, log_split, log_degree, p)


@pytest.mark.gpu
@pytest.mark.parametrize("degree,A,B", [(14, -6.0, 6.0), (62, -10.0, 10.0)])
def test_sigmoid_approximation_change_of_variable_and_evaluate_cheby(tmp_path, degree, A, B):
    """mpc.CSigmoidApprox's local computation (mhe.go:634-667; Degree = 62 on [-10, 10] is configGlobal.toml's default) on device vectors, real keys:
    (a) words, level and scale equal an oracle replay of every evaluator step of the restated EvaluateCheby (unpinned against lattigo itself);
    (b) the result decrypts to the sigmoid within the interpolation error; (c) the level budget is the documented ceil(log2(deg + 1)) + 1 (+ 1 for the
    change of variable), with the leading branch evaluated one level higher (the re-split)."""
    import pyref
    from sfgwas_amd import capi
    capi.lib()
    ol.build_oracle()
    exe = build("host_sigmoid_test", with_oracle=True)
    # A chain with q_i ~ Delta (34-bit primes under the 2^34 scale, as PN15 / PN16 pair 40 / 45-bit primes with their scales): every MulRelin + Rescale divides
    # by exactly one modulus.  On PN14QP438 itself (35-bit primes, 2^34 scale) lattigo's Rescale rule skips the division whenever q > 2^35 (T_2 would keep scale
    # 2^68 and its leaf constant one bit), so EvaluateCheby is not meaningful there and the mirror refuses ("scale out of range") instead of returning noise.
    Q, Pp = ol.small_primes(14, 46, 1) + ol.small_primes(14, 34, 9), ol.small_primes(14, 43, 2)
    ring = ol.Ring(14, Q, Pp)
    slots, SC, top, seed, nct = 8192, 2.0 ** 34, 9, 4242, 2
    sec = ring.gen_secret(seed)
    rlk = np.zeros(ring.key_words(), dtype=np.uint64)
    ol.lib().orc_gen_rlk(ring.h, ol.pi8(sec), 4999, ol.p64(rlk))
    np.array([len(Q), len(Pp)] + Q + Pp, dtype=np.uint64).tofile(tmp_path / "moduli.bin")
    (tmp_path / "case.txt").write_text(f"{nct} {top} {degree} {A} {B} {seed}\n")
    rnd = np.random.default_rng(degree)
    xs = rnd.uniform(0.9 * A, 0.9 * B, (nct, slots))
    X = np.stack([ring.encrypt(sec, top, ring.encode_coeffs(xs[k], SC), 700 + k) for k in range(nct)])
    X.tofile(tmp_path / "x.bin")
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-3000:]
    coeffs = np.fromfile(tmp_path / "coeffs.bin", dtype=np.float64)
    # ckks.Approximate, recomputed: the nodes' cosines may differ in the last bit between libm and numpy, the coefficients by ~1e-16
    n = degree + 1
    nodes = 0.5 * (A + B) + 0.5 * (B - A) * np.cos((np.arange(1, n + 1) - 0.5) * (np.pi / n))
    u = (2 * nodes - A - B) / (B - A)
    Tm = np.polynomial.chebyshev.chebvander(u, degree)
    mine = (1.0 / (1 + np.exp(-nodes))) @ Tm * (2.0 / n); mine[0] /= 2
    assert coeffs.shape == (n,) and np.max(np.abs(coeffs - mine)) < 1e-13
    n_out, lvl, sc = (tmp_path / "meta.txt").read_text().split()
    got = np.fromfile(tmp_path / "y.bin", dtype=np.uint64).reshape(nct, 2, int(lvl) + 1, ring.N)
    depth = int(np.ceil(np.log2(degree + 1))) + 1
    assert top - 1 - int(lvl) <= depth, f"EvaluateCheby used {top - 1 - int(lvl)} levels, documented bound {depth}"
    for k in range(nct):
        x = _Ct(X[k], top, SC)
        y = _r_mul_const(ring, x, 2 / (B - A))
        y = _r_rescale_once(ring, y, SC)
        y = _r_add_const(ring, y, (-A - B) / (B - A))
        rep = _ChebyReplay(ring, rlk, SC, y)
        w = rep.evaluate(list(coeffs), y.scale)
        assert (int(lvl), float(sc)) == (w.level, w.scale), f"level / scale ({lvl}, {sc}) vs the replay's ({w.level}, {w.scale!r})"
        assert np.array_equal(got[k], w.a), f"ciphertext {k}: words"
        res = ring.decrypt_residues(sec, w.level, got[k])
        nm = min(4, w.level + 1)
        big = pyref.crt_centered([res[m] for m in range(nm)], ring.moduli[:nm])
        dec = pyref.decode(np.array([float(v) for v in big]) / w.scale, ring.N).real
        ref = 1.0 / (1 + np.exp(-xs[k]))
        cheb = np.polynomial.chebyshev.chebval((2 * xs[k] - A - B) / (B - A), coeffs)
        assert np.max(np.abs(dec - cheb)) < 2e-4, f"decrypted values are off the interpolant by {np.max(np.abs(dec - cheb))}"
        assert np.max(np.abs(dec - ref)) < (2e-2 if degree < 20 else 2e-3), f"decrypted values are off the sigmoid by {np.max(np.abs(dec - ref))}"


@pytest.mark.gpu
def test_host_mirror_on_a_multi_gpu_party_matches_the_oracle(tmp_path):
    """crypto::NewCryptoParamsMulti({0, 0, 0}) - three ranks of the library's multi-GPU engine on one device (in-process direct transport) - behind the unchanged
    call sites of pca.go:112-113,344,352: Preprocess(X), Preprocess(X^T as the transposed view), Compute(Q X), Compute(Q' X^T) on a ragged 1 x 3 block matrix, every
    word against the oracle (gwas/matmult.go:1043-1236)."""
    from sfgwas_amd import capi
    capi.lib()
    exe = build("host_mgpu_test")
    ring = ol.Ring(14, ol.Q_PN14, ol.P_PN14)
    keys = ol.RotKeys(ring)
    rnd = np.random.default_rng(41)
    slots, d = 8192, 91
    nrow, ncol, s, level, L = 70, 2 * slots + 40, 2, 5, 5
    geno = rnd.integers(-1, 3, (nrow, ncol)).astype(np.int8)
    geno.tofile(tmp_path / "geno.bin")
    np.ascontiguousarray(geno.T).tofile(tmp_path / "geno_t.bin")                                  # pca.go:113 registers X^T from its own file (gwas.go:597)
    other = np.ascontiguousarray(geno.T); other[ncol - 1, nrow - 1] ^= 1; other.tofile(tmp_path / "geno_u.bin")     # X^T's shape, one entry different
    steps = set(range(1, d)) | {g * d for g in range(1, d) if g * d < slots} | {slots - 1}       # every baby and giant step (the engine's rank-local rotation cache rotates all 91 babies)
    blob = [np.array([len(steps)], dtype=np.uint64)]
    for k in sorted(steps):
        g = ring.galois(k)
        key = capi.random_rotkey(ring.moduli, ring.beta, ring.N, 300 + k)
        keys.add(g, key)
        blob += [np.array([g], dtype=np.uint64), key.reshape(-1)]
    np.concatenate(blob).tofile(tmp_path / "keys.bin")
    np.array([len(ol.Q_PN14), len(ol.P_PN14)] + ol.Q_PN14 + ol.P_PN14, dtype=np.uint64).tofile(tmp_path / "moduli.bin")
    A = np.stack([np.stack([ring.fill_uniform(level, 50 + i)]) for i in range(s)])                            # Q : s x 1 block row
    AT = np.stack([np.stack([ring.fill_uniform(level, 70 + 3 * i + b) for b in range(3)]) for i in range(s)])     # Q': s x 3 SNP blocks
    A.tofile(tmp_path / "A.bin"); AT.tofile(tmp_path / "AT.bin")
    (tmp_path / "case.txt").write_text(f"{nrow} {ncol} {s} {level} {L} 0\n")
    out = subprocess.run([exe, str(tmp_path), "0,0,0"], capture_output=True, text=True)
    assert out.returncode == 0 and "OK direct world 3" in out.stdout, (out.stdout, out.stderr)
    want, _, _ = ol.matmult4stream(ring, keys, 2.0 ** 34, A, level, L, geno)
    assert np.array_equal(np.fromfile(tmp_path / "out_x.bin", dtype=np.uint64).reshape(want.shape), want)
    want_t, _, _ = ol.matmult4stream(ring, keys, 2.0 ** 34, AT, level, L, np.ascontiguousarray(geno.T))
    assert np.array_equal(np.fromfile(tmp_path / "out_xt.bin", dtype=np.uint64).reshape(want_t.shape), want_t)
    rot = np.fromfile(tmp_path / "rot.bin", dtype=np.uint64).reshape(A[0, 0].shape)
    assert np.array_equal(rot, ol.rotate_right(ring, keys, level, A[0, 0], 1))
