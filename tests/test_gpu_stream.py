"""Association batches streamed from a PLINK .bed on disk (gwas/assoc.go:340-420 GenoBlockMult): batching by kept SNPs, row / column
filters, per-batch MatMult4Stream, ConcatCipherMatrix layout, padded column sums - every output word vs the oracle run on the batch matrices
that numpy decodes from the same file."""
import ctypes as C
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu


def write_bed(path, geno):
    """geno: [num_sample][num_snp] int8 in {2, -1, 1, 0} -> SNP-major 2-bit codes {2: 00, -1: 01, 1: 10, 0: 11} (scripts/plinkBedToBinary.py:18-27)"""
    code = {2: 0, -1: 1, 1: 2, 0: 3}
    ns, nv = geno.shape
    bps = (ns + 3) // 4
    out = bytearray([0x6C, 0x1B, 0x01])
    for j in range(nv):
        col = bytearray(bps)
        for i in range(ns):
            col[i // 4] |= code[int(geno[i, j])] << (2 * (i % 4))
        out += col
    open(path, "wb").write(bytes(out))


def batches(colf, batch):
    """assoc.go:371-416"""
    out, start, counter = [], 0, 0
    for idx in range(len(colf)):
        counter += int(colf[idx])
        if counter == batch or (idx == len(colf) - 1 and counter > 0):
            out.append((start, idx + 1)); start, counter = idx + 1, 0
    return out


@pytest.mark.parametrize("square", [False, True])
def test_stream_bed_batches_match_oracle(tmp_path, square):
    from sfgwas_amd import capi
    ns, nv, batch, s, level, maxl = 130, 260, 100, 2, 5, 5
    rnd = np.random.default_rng(17)
    geno = rnd.choice(np.array([2, -1, 1, 0], dtype=np.int8), size=(ns, nv), p=[0.2, 0.05, 0.35, 0.4])
    rowf = (rnd.random(ns) < 0.9).astype(np.uint8); colf = (rnd.random(nv) < 0.85).astype(np.uint8)
    path = str(tmp_path / "chr1.bed")
    write_bed(path, geno)
    ctx = capi.Context(ol.Q_PN14, ol.P_PN14)
    ring = ol.Ring(14, ol.Q_PN14, ol.P_PN14)
    keys = ol.RotKeys(ring)
    nr = int(rowf.sum())
    slots, d = ring.slots, 91
    shifts = set(range(nr)) | set(range(slots - batch + 1, slots))
    rots = sorted({sh % d for sh in shifts if sh % d} | {(sh // d) * d for sh in shifts if sh // d})
    for k in rots:
        g = ring.galois(k)
        key = capi.random_rotkey(ring.moduli, ring.beta, ring.N, 900 + k)
        keys.add(g, key); ctx.load_rotkey(g, key)
    A = np.stack([np.stack([ring.fill_uniform(level, 40 + i)]) for i in range(s)])          # [s][1 block row]
    bt = batches(colf, batch)
    nct_total = sum((int(colf[a:b].sum()) - 1) // slots + 1 for a, b in bt)
    cap = nct_total + 1
    dA = capi.DevArray.from_host(ctx, A)
    dout = capi.DevArray(ctx, (s, cap, 2, maxl, ring.N))
    sums = np.full(cap * slots, -7.0); sq = np.full(cap * slots, -7.0)
    got_ct = C.c_size_t()
    flags = capi.SFG_SQUARE if square else 0
    ctx.check(capi.lib().sfg_assoc_stream_bed(ctx.h, path.encode(), ns, nv, rowf.ctypes.data_as(C.c_void_p), colf.ctypes.data_as(C.c_void_p), batch,
                                              dA.p, s, level, maxl, flags, dout.p, cap, C.byref(got_ct), sums.ctypes.data_as(C.c_void_p),
                                              sq.ctypes.data_as(C.c_void_p)), "assoc_stream_bed")
    assert got_ct.value == nct_total and len(bt) == 3
    out = dout.host()
    shift = 0
    for a, b in bt:
        sub = np.ascontiguousarray(geno[rowf.astype(bool)][:, a:b][:, colf[a:b].astype(bool)])
        want, wsum, wsq = ol.matmult4stream(ring, keys, 2.0 ** 34, A, level, maxl, sub, compute_sqsum=True, square=square, enc_prec=1)
        nct = want.shape[1]
        assert np.array_equal(out[:, shift:shift + nct], want), f"batch {a}:{b}"
        assert np.array_equal(sums[shift * slots: shift * slots + sub.shape[1]], wsum) and np.array_equal(sq[shift * slots: shift * slots + sub.shape[1]], wsq)
        assert not sums[shift * slots + sub.shape[1]: (shift + nct) * slots].any()               # padded tail of the reference's dosage vectors
        shift += nct
    assert (sums[shift * slots:] == -7.0).all()
    # errors the reference would panic on
    with pytest.raises(capi.SfgError, match="cannot open"):
        ctx.check(capi.lib().sfg_assoc_stream_bed(ctx.h, b"/nonexistent.bed", ns, nv, None, None, batch, dA.p, s, level, maxl, 0, dout.p, cap, C.byref(got_ct), None, None), "x")
    with pytest.raises(capi.SfgError, match="expected 3 \\+"):
        ctx.check(capi.lib().sfg_assoc_stream_bed(ctx.h, path.encode(), ns + 4, nv, None, None, batch, dA.p, s, level, maxl, 0, dout.p, cap, C.byref(got_ct), None, None), "x")
    dA.free(); dout.free(); ctx.close()


@pytest.mark.parametrize("s,level,maxl,form", [(16, 5, 5, "assoc.rotf"), (3, 5, 3, "assoc.rot8"), (15, 4, 4, "assoc.rot8")])
def test_stream_bed_cache_forms_at_their_limits(tmp_path, s, level, maxl, form):
    """The call-wide rotation cache is kept as int8 rot tiles while the 2 s ciphertext rows of a k-slice fit the MAC's two row tiles in one launch (s <= 15) and as fp64 operand
    rows beyond (s = 16); a product below the top level (max_level 3 and 4: the 46-bit modulus and two or three 35-bit ones) takes the tiles of those moduli only.
    Two batches each, every word against the oracle."""
    from sfgwas_amd import capi
    ns, nv, batch = 120, 150, 80
    rnd = np.random.default_rng(1000 + s)
    geno = rnd.choice(np.array([2, -1, 1, 0], dtype=np.int8), size=(ns, nv), p=[0.2, 0.05, 0.35, 0.4])
    path = str(tmp_path / "lim.bed")
    write_bed(path, geno)
    ctx = capi.Context(ol.Q_PN14, ol.P_PN14)
    ring = ol.Ring(14, ol.Q_PN14, ol.P_PN14)
    keys = ol.RotKeys(ring)
    slots, d = ring.slots, 91
    shifts = set(range(ns)) | set(range(slots - batch + 1, slots))
    rots = sorted({sh % d for sh in shifts if sh % d} | {(sh // d) * d for sh in shifts if sh // d})
    for k in rots:
        g = ring.galois(k)
        key = capi.random_rotkey(ring.moduli, ring.beta, ring.N, 700 + k)
        keys.add(g, key); ctx.load_rotkey(g, key)
    A = np.stack([np.stack([ring.fill_uniform(level, 300 + i)]) for i in range(s)])
    dA = capi.DevArray.from_host(ctx, A)
    dout = capi.DevArray(ctx, (s, 2, 2, maxl, ring.N))
    got_ct = C.c_size_t()
    ctx.check(capi.lib().sfg_assoc_stream_bed(ctx.h, path.encode(), ns, nv, None, None, batch, dA.p, s, level, maxl, 0, dout.p, 2, C.byref(got_ct), None, None), "assoc_stream_bed")
    assert got_ct.value == 2
    for name in (b"assoc.rot8", b"assoc.rotf"):
        n = C.c_size_t()
        ctx.check(capi.lib().sfg_ctx_scratch_bytes(ctx.h, name, C.byref(n)), "scratch_bytes")
        assert (n.value > 0) == (name.decode() == form), (name, n.value)
    out = dout.host()
    for b, (a0, a1) in enumerate([(0, batch), (batch, nv)]):
        want, _, _ = ol.matmult4stream(ring, keys, 2.0 ** 34, A, level, maxl, np.ascontiguousarray(geno[:, a0:a1]), enc_prec=1)
        assert np.array_equal(out[:, b:b + 1], want), f"batch {b}"
    dA.free(); dout.free(); ctx.close()


def _ctx_with_env(capi, **env):
    """the library reads its switches once, in sfg_ctx_create"""
    import os
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update({k: str(v) for k, v in env.items()})
    try:
        return capi.Context(ol.Q_PN14, ol.P_PN14)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_gwy_four_call_sequence_over_sixteen_streamed_batches(tmp_path):
    """The logistic path's gWY (assoc.go:1338,1375,1404,1422) calls GenoBlockMult four times per block with s = ncov, 1 (square = true), 1, 1 on the SAME
    batches.  16 batches of 64 kept SNPs streamed from one .bed: every call with the call-wide baby-step rotation cache as int8 rot tiles (the default),
    without it (SFG_ASSOC_ROTCACHE_MB=0: the reference's per-batch rebuild) and, where the file system allows, with O_DIRECT reads - identical words; batch 5 of every
    call against the oracle.  (The cache as fp64 operand rows - round 3's form - lives in the A/B build only since round 6.)"""
    from sfgwas_amd import capi
    ns, nv, batch, level, maxl, ncov = 300, 1100, 64, 5, 5, 5
    rnd = np.random.default_rng(23)
    geno = rnd.choice(np.array([2, -1, 1, 0], dtype=np.int8), size=(ns, nv), p=[0.2, 0.05, 0.35, 0.4])
    colf = (rnd.random(nv) < 0.95).astype(np.uint8)
    path = str(tmp_path / "chr2.bed")
    write_bed(path, geno)
    ring = ol.Ring(14, ol.Q_PN14, ol.P_PN14)
    keys = ol.RotKeys(ring)
    slots, d = ring.slots, 91
    shifts = set(range(ns)) | set(range(slots - batch + 1, slots))
    rots = sorted({sh % d for sh in shifts if sh % d} | {(sh // d) * d for sh in shifts if sh // d})
    ctxs = {"cached": capi.Context(ol.Q_PN14, ol.P_PN14), "per_batch": _ctx_with_env(capi, SFG_ASSOC_ROTCACHE_MB=0)}
    for k in rots:
        g = ring.galois(k)
        key = capi.random_rotkey(ring.moduli, ring.beta, ring.N, 900 + k)
        keys.add(g, key)
        for c in ctxs.values():
            c.load_rotkey(g, key)
    bt = batches(colf, batch)
    assert len(bt) >= 16
    cap = len(bt)
    calls = [(ncov, False, 61), (1, True, 62), (1, False, 63), (1, False, 64)]            # (s, square, seed): WzBT, w (squared genotypes), yTilde, WzZTwZInvZTy
    for s, square, seed in calls:
        A = np.stack([np.stack([ring.fill_uniform(level, seed * 10 + i)]) for i in range(s)])
        outs = {}
        for name, ctx in ctxs.items():
            for direct in ((False, True) if name == "cached" else (False,)):
                dA = capi.DevArray.from_host(ctx, A)
                dout = capi.DevArray(ctx, (s, cap, 2, maxl, ring.N))
                got_ct = C.c_size_t()
                flags = (capi.SFG_SQUARE if square else 0) | (capi.SFG_STREAM_DIRECT if direct else 0)
                rc = capi.lib().sfg_assoc_stream_bed(ctx.h, path.encode(), ns, nv, None, colf.ctypes.data_as(C.c_void_p), batch, dA.p, s, level, maxl, flags,
                                                     dout.p, cap, C.byref(got_ct), None, None)
                if rc and direct and b"O_DIRECT" in capi.lib().sfg_last_error(ctx.h):
                    dA.free(); dout.free()
                    continue                                                              # tmpfs and friends: no O_DIRECT
                ctx.check(rc, "assoc_stream_bed")
                assert got_ct.value == len(bt)
                outs[(name, direct)] = dout.host()
                dA.free(); dout.free()
        # the call-wide cache is kept as the int8 MAC's rot tiles (default) or not at all
        kept = {}
        for name, ctx in ctxs.items():
            n8, nf = C.c_size_t(), C.c_size_t()
            capi.lib().sfg_ctx_scratch_bytes(ctx.h, b"assoc.rot8", C.byref(n8)); capi.lib().sfg_ctx_scratch_bytes(ctx.h, b"assoc.rotf", C.byref(nf))
            kept[name] = (n8.value > 0, nf.value > 0)
        assert kept == {"cached": (True, False), "per_batch": (False, False)}, kept
        ref = outs[("cached", False)]
        for key_, o in outs.items():
            assert np.array_equal(o, ref), f"s={s} square={square}: {key_} differs from the cached path"
        a, b = bt[5]
        sub = np.ascontiguousarray(geno[:, a:b][:, colf[a:b].astype(bool)])
        want, _, _ = ol.matmult4stream(ring, keys, 2.0 ** 34, A, level, maxl, sub, square=square, enc_prec=1)
        assert np.array_equal(ref[:, 5:6], want), f"s={s} square={square}: batch 5 vs the oracle"
    for c in ctxs.values():
        c.close()
