"""Genotype input formats, pinned by outputs of the reference's OWN converters (tests/golden/input_formats.npz was
produced by running scripts/plinkBedToBinary.py, filterMatrix.py, transposeMatrix.py, mergeMatrices.py; see
tests/golden/make_input_fixtures.py).  CPU: the oracle restatement vs the fixtures.  GPU: sfg_geno_from_bed /
transpose / concat_cols vs the fixtures, bit-exact."""
import ctypes as C
import os
import numpy as np
import pytest

import oracle_lib as ol

FX = np.load(os.path.join(os.path.dirname(__file__), "golden", "input_formats.npz"))
CASES = [tuple(int(x) for x in c) for c in FX["bed_cases"]]


def vp(a):
    return a.ctypes.data_as(C.c_void_p)


@pytest.mark.parametrize("k", range(len(CASES)))
def test_oracle_bed_decode_and_filter_match_reference_scripts(k):
    ns, nv = CASES[k]
    bed = np.ascontiguousarray(FX[f"bed_{k}"])
    out = np.empty((ns, nv), dtype=np.int8)
    assert ol.lib().orc_bed_decode(vp(bed), bed.size, ns, nv, vp(out)) == 0
    assert np.array_equal(out, FX[f"geno_{k}"])
    rf, cf = np.ascontiguousarray(FX[f"rowfilt_{k}"]), np.ascontiguousarray(FX[f"colfilt_{k}"])
    filt = np.empty((int(rf.sum()), int(cf.sum())), dtype=np.int8)
    ol.lib().orc_filter_matrix(vp(out), ns, nv, vp(rf), vp(cf), vp(filt))
    assert np.array_equal(filt, FX[f"filtered_{k}"])
    assert ol.lib().orc_bed_decode(vp(bed), bed.size - 1, ns, nv, vp(out)) != 0      # the script's length assert


@pytest.fixture(scope="module")
def ctx():
    from sfgwas_amd import capi
    c = capi.Context(ol.Q_PN14, ol.P_PN14)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("k", range(len(CASES)))
def test_gpu_bed_decode_filter_transpose_match_reference_scripts(ctx, k):
    from sfgwas_amd import capi
    ns, nv = CASES[k]
    bed = FX[f"bed_{k}"]
    g = ctx.geno_from_bed(bed, ns, nv)
    assert np.array_equal(ctx.geno_to_host(g), FX[f"geno_{k}"])
    gt = C.c_void_p()
    ctx.check(capi.lib().sfg_geno_transpose(ctx.h, g, C.byref(gt)), "transpose")
    assert np.array_equal(ctx.geno_to_host(gt), FX[f"transposed_{k}"])
    gf = ctx.geno_from_bed(bed, ns, nv, FX[f"rowfilt_{k}"], FX[f"colfilt_{k}"])
    assert np.array_equal(ctx.geno_to_host(gf), FX[f"filtered_{k}"])
    for h in (g, gt, gf):
        capi.lib().sfg_geno_free(ctx.h, h)


@pytest.mark.gpu
def test_gpu_bed_errors_and_merge(ctx):
    from sfgwas_amd import capi
    ns, nv = CASES[0]
    bed = FX["bed_0"].copy()
    with pytest.raises(capi.SfgError, match="expected 3 \\+"):
        ctx.geno_from_bed(bed[:-1], ns, nv)
    bad = bed.copy(); bad[2] = 0                                   # sample-major .bed is not what the script reads
    with pytest.raises(capi.SfgError, match="not a SNP-major"):
        ctx.geno_from_bed(bad, ns, nv)
    with pytest.raises(capi.SfgError, match="keep nothing"):
        ctx.geno_from_bed(bed, ns, nv, np.zeros(ns, dtype=np.uint8), None)
    parts = [np.ascontiguousarray(FX[f"merge_part_{i}"]) for i in range(3)]
    hs = []
    for p in parts:
        h = C.c_void_p()
        ctx.check(capi.lib().sfg_geno_upload(ctx.h, vp(p), p.shape[0], p.shape[1], p.shape[1], C.byref(h)), "upload")
        hs.append(h)
    arr = (C.c_void_p * 3)(*[h.value for h in hs])
    m = C.c_void_p()
    ctx.check(capi.lib().sfg_geno_concat_cols(ctx.h, arr, 3, C.byref(m)), "concat")
    assert np.array_equal(ctx.geno_to_host(m), FX["merged"])
    for h in hs + [m]:
        capi.lib().sfg_geno_free(ctx.h, h)


@pytest.mark.gpu
def test_gpu_bed_decode_large_roundtrip(ctx):
    """size-independent property at a multi-tile size: decode(pack(G)) == G, and a product fed from the decoded
    handle equals one fed from the uploaded int8 matrix"""
    from sfgwas_amd import capi
    rnd = np.random.default_rng(5)
    ns, nv = 4099, 2051
    G = rnd.integers(-1, 3, (ns, nv)).astype(np.int8)
    code = np.zeros_like(G, dtype=np.uint8)
    code[G == 2] = 0; code[G == -1] = 1; code[G == 1] = 2; code[G == 0] = 3
    bps = (ns + 3) // 4
    pad = np.zeros((bps * 4, nv), dtype=np.uint8); pad[:ns] = code
    packed = (pad[0::4] | (pad[1::4] << 2) | (pad[2::4] << 4) | (pad[3::4] << 6)).T.copy()     # [nv][bps]
    bed = np.concatenate([np.array([0x6C, 0x1B, 0x01], dtype=np.uint8), packed.reshape(-1)])
    g = ctx.geno_from_bed(bed, ns, nv)
    assert np.array_equal(ctx.geno_to_host(g), G)
    capi.lib().sfg_geno_free(ctx.h, g)
