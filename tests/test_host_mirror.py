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
        dct = _orc_sub(ring, 4, prod[i, 0], _drop(qsm, 4))
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
        dct = _orc_sub(ring, 4, prod2[i, 0], _drop(q1m, 4))
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
def _orc_cmult_vec(ring, level, scale_a, scale_b, A, B, rlk):
    """crypto.CMult on vectors with length-1 broadcast: list of (ct, level, scale)"""
    n = max(len(A), len(B))
    return [_orc_cmult(ring, level, scale_a * scale_b, A[k % len(A)], B[k % len(B)], rlk) for k in range(n)]


def _bootstrap1(ring, level, ct, ct_scale, sk, mask, e0, e1, crs):
    """one party: its own shares are the aggregate (mhe.go:313-331 at the target scale Params.Scale())"""
    h0, h1 = ol.refresh_gen_shares_scaled(ring, level, ct, ct_scale, 2.0 ** 34, sk, crs, mask, e0, e1)
    return ol.refresh_finish_scaled(ring, level, ct, ct_scale, 2.0 ** 34, h0, h1, crs)


@pytest.mark.gpu
def test_power_iteration_local_segments_with_bootstraps_at_two_block_rows_and_columns(tmp_path):
    """pca.go:339-353 local work at n_ind = 8192 + 45, m_snp = 8192 + 33 (nbr = m_ct = 2, ragged second blocks), kp = 2, REAL keys from a toy secret:
    QXtLazyNormStream part 1 -> bootstrap (levels 4 -> 9, scale A.scale * Delta -> Delta) -> part 2 -> bootstrap -> QXLazyNormStream part 1 -> bootstrap
    -> part 2 with MaskTrunc on the ragged column only: the result carries per-ciphertext levels and scales (matmult.go:60-70).
    (a) every dumped word equals the oracle's replay of the same composition; (b) the final cells DECRYPT, at their own scales, to
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
    # plaintext inputs: Q kp x n_ind, SNP means / inverse standard deviations
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
    rA = randomness("bootA", s * mct, 4)
    lvlA_out = 7                                     # Q1m = CMultScalar at 9 -> 8; Sub at 8; CMult with XStdInv + rescale -> 7
    rM = randomness("bootM", s * mct, lvlA_out)
    rB = randomness("bootB", s * nbr, 4)
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-3000:]
    meta = {ln.split()[0]: ln.split()[1:] for ln in (tmp_path / "meta.txt").read_text().splitlines()}

    def ld(name):
        r, c, lvl, sc = int(meta[name][0]), int(meta[name][1]), int(meta[name][2]), float(meta[name][3])
        return np.fromfile(tmp_path / (name + ".bin"), dtype=np.uint64).reshape(r, c, 2, lvl + 1, ring.N), lvl, sc
    # ---------------- oracle replay, A
    prodA, _, _ = ol.matmult4stream(ring, keys, SC, Q, top, 5, geno, enc_prec=1)
    got, lvl, sc = ld("a_prod")
    assert lvl == 4 and sc == SC * SC and np.array_equal(got, prodA), "A: product"
    bootA = np.stack([np.stack([_bootstrap1(ring, 4, prodA[i, j], SC * SC, sk, rA[0][i * mct + j], rA[2][i * mct + j], rA[3][i * mct + j], rA[1][i * mct + j])
                                for j in range(mct)]) for i in range(s)])
    got, lvl, sc = ld("a_boot")
    assert lvl == top and sc == SC and np.array_equal(got, bootA), "A: bootstrap at the target scale"
    outA, scA = [], None
    for i in range(s):
        row_sum = _orc_innersum(ring, keys, top, list(Q[i]))
        q1m = _orc_cmult_vec(ring, top, SC, SC, list(XMean), [row_sum], rlk)
        l1 = q1m[0][1]
        dct = [_orc_sub(ring, l1, _drop(bootA[i, j], l1), q1m[j][0]) for j in range(mct)]
        res = _orc_cmult_vec(ring, l1, SC, SC, dct, [_drop(x, l1) for x in XStdInv], rlk)
        outA.append(np.stack([r[0] for r in res])); lA, scA = res[0][1], res[0][2]
    outA = np.stack(outA)
    got, lvl, sc = ld("a_out")
    assert lvl == lA == lvlA_out and sc == scA and np.array_equal(got, outA), "A: lazy normalisation after the bootstrap"
    Q1 = np.stack([np.stack([_bootstrap1(ring, lA, outA[i, j], scA, sk, rM[0][i * mct + j], rM[2][i * mct + j], rM[3][i * mct + j], rM[1][i * mct + j])
                             for j in range(mct)]) for i in range(s)])
    got, lvl, sc = ld("q1")
    assert lvl == top and sc == SC and np.array_equal(got, Q1), "bootstrap between the two products (scale not a power of two -> Delta)"
    # ---------------- oracle replay, B
    QS = [_orc_cmult_vec(ring, top, SC, SC, list(Q1[i]), list(XStdInv), rlk) for i in range(s)]
    lQS, scQS = QS[0][0][1], QS[0][0][2]
    QSm = np.stack([np.stack([c[0] for c in QS[i]]) for i in range(s)])
    prodB, _, _ = ol.matmult4stream(ring, keys, SC, QSm, lQS, 5, np.ascontiguousarray(geno.T), enc_prec=1)
    got, lvl, sc = ld("b_prod")
    assert lvl == 4 and sc == scQS * SC and np.array_equal(got, prodB), "B: product"
    bootB = np.stack([np.stack([_bootstrap1(ring, 4, prodB[i, j], scQS * SC, sk, rB[0][i * nbr + j], rB[2][i * nbr + j], rB[3][i * nbr + j], rB[1][i * nbr + j])
                                for j in range(nbr)]) for i in range(s)])
    got, lvl, sc = ld("b_boot")
    assert lvl == top and sc == SC and np.array_equal(got, bootB), "B: bootstrap"
    mask = np.zeros(slots); mask[:((n_ind - 1) % slots) + 1] = 1.0
    # plaintext of the whole chain
    R1 = (Qp @ G - np.outer(Qp.sum(1), mean)) * sinv
    QSp = R1 * sinv
    R2 = QSp @ G.T - np.outer(QSp @ mean, np.ones(n_ind))
    for i in range(s):
        prods = _orc_cmult_vec(ring, lQS, scQS, SC, [c[0] for c in QS[i]], [_drop(x, lQS) for x in XMean], rlk)
        l2, s2 = prods[0][1], prods[0][2]
        qsm = _orc_innersum(ring, keys, l2, [p[0] for p in prods])
        for j in range(nbr):
            dct = _orc_sub(ring, l2, _drop(bootB[i, j], l2), qsm)
            name = f"b_out_{i}_{j}"
            got, lvl, sc = ld(name)
            if j + 1 < nbr:                                      # full-slot column: MaskTrunc returns it untouched
                want, wl, ws = dct, l2, SC
            else:
                mp = np.zeros_like(dct)
                mpt = ring.encode_ntt(mask, SC, l2 + 1)
                ol.lib().orc_mul_plain(ring.h, l2, ol.p64(dct), ol.p64(mpt), ol.p64(mp))
                want, wl, ws = _orc_rescale_loop(ring, mp, l2, SC * SC)
            assert lvl == wl and sc == ws, f"{name}: level / scale bookkeeping ({lvl}, {sc}) vs ({wl}, {ws})"
            assert np.array_equal(got[0, 0], want), f"{name}: words"
            # decode-level check at the ciphertext's OWN scale
            res = ring.decrypt_residues(sec, lvl, got[0, 0])
            big = pyref.crt_centered([res[m] for m in range(3)], ring.moduli[:3])
            dec = pyref.decode(np.array([float(x) for x in big]) / sc, ring.N).real
            ref = np.zeros(slots); seg = R2[i, j * slots:(j + 1) * slots]; ref[:len(seg)] = seg
            assert np.max(np.abs(dec - ref)) < 2e-3 * max(1.0, np.max(np.abs(R2))), f"{name}: decrypted values off by {np.max(np.abs(dec - ref))}"
    assert int(meta["b_out_0_0"][2]) == int(meta["b_out_0_1"][2]) + 1, "the full column must stay one level above the masked tail"
