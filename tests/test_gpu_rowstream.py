"""Row-streamed registration of the resident matrix.  The reference never holds its matrix: MatMult4StreamPreprocess (gwas/matmult.go:914-1041) pulls one row at a time
out of GenoFileStream.NextRow (gwas/filestream.go:414-426), and gwas/pca.go:112-113 registers X and then X^T from two files.  sfg_geno_create / _write_rows fill the
resident matrix chunk by chunk; sfg_geno_compare_rows recognises the second file as the transpose of the first ON the device (exact comparison), so that one int8 copy
serves both products.  Checked here: chunked = whole upload (bytes and product words vs the oracle), ragged last chunk, strided staging buffer, the multi-GPU scatter
(3 ranks on one device), transposed-view recognition and its negative case, argument errors."""
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
T = 2


def chunks(n, step):
    return [(lo, min(lo + step, n)) for lo in range(0, n, step)]


def test_chunked_upload_is_the_whole_upload_and_its_product_matches_the_oracle():
    from sfgwas_amd import capi
    ctx = capi.Context(ol.Q_PN14, ol.P_PN14)
    ring = ol.Ring(14, ol.Q_PN14, ol.P_PN14)
    keys = ol.RotKeys(ring)
    rnd = np.random.default_rng(61)
    nrow, ncol, s, level = 97, 53, 1, 5
    geno = rnd.integers(-1, 3, (nrow, ncol)).astype(np.int8)
    d, slots = 91, 8192
    shifts = set(range(nrow)) | set(range(slots - ncol + 1, slots))
    for k in sorted({sh % d for sh in shifts if sh % d} | {(sh // d) * d for sh in shifts if sh // d}):
        key = capi.random_rotkey(ring.moduli, ring.beta, ring.N, 900 + k)
        keys.add(ring.galois(k), key); ctx.load_rotkey(ring.galois(k), key)
    g = ctx.geno_create(nrow, ncol)
    staging = np.full((40, ncol + 11), 99, dtype=np.int8)                 # a reused staging buffer wider than the matrix: row stride ld > ncol
    for lo, hi in chunks(nrow, 40):                                       # 40 + 40 + 17: ragged last chunk
        staging[:hi - lo, :ncol] = geno[lo:hi]
        ctx.check(capi.lib().sfg_geno_write_rows(ctx.h, g, lo, hi - lo, staging.ctypes.data, staging.shape[1]), "write_rows")
    back = np.zeros((nrow, ncol), dtype=np.int8)
    ctx.check(capi.lib().sfg_geno_download(ctx.h, g, back.ctypes.data), "download")
    assert np.array_equal(back, geno)
    A = np.stack([np.stack([ring.fill_uniform(level, 5)])])
    dA = capi.DevArray.from_host(ctx, A)
    out = ctx.matmul_resident(dA, s, level, 5, g, 0)
    want, _, _ = ol.matmult4stream(ring, keys, 2.0 ** 34, A, level, 5, geno, enc_prec=1)
    assert np.array_equal(out.host().reshape(want.shape), want)
    # the same matrix arriving again, and its transpose arriving row by row (pca.go:113), are recognised on the device; one changed entry is not
    assert sum(ctx.geno_compare_rows(g, 0, lo, geno[lo:hi]) for lo, hi in chunks(nrow, 33)) == 0
    gt = np.ascontiguousarray(geno.T)
    assert sum(ctx.geno_compare_rows(g, T, lo, gt[lo:hi]) for lo, hi in chunks(ncol, 20)) == 0
    gt2 = gt.copy(); gt2[ncol - 1, nrow - 1] ^= 1; gt2[0, 0] ^= 3
    assert sum(ctx.geno_compare_rows(g, T, lo, gt2[lo:hi]) for lo, hi in chunks(ncol, 20)) == 2
    assert ctx.geno_compare_rows(g, 0, 3, np.ascontiguousarray(geno[4:9])) > 0       # right rows, wrong place
    with pytest.raises(capi.SfgError, match="rows"):
        ctx.geno_write_rows(g, nrow - 2, geno[:5])
    with pytest.raises(capi.SfgError, match="rows"):
        ctx.geno_compare_rows(g, T, ncol - 1, gt[:2])
    view = capi.C.c_void_p()                                              # a non-owning view (sfg_geno_from_device) is not written into
    ctx.check(capi.lib().sfg_geno_from_device(ctx.h, back.ctypes.data, nrow, ncol, ncol, capi.C.byref(view)), "from_device")     # (never dereferenced)
    with pytest.raises(capi.SfgError, match="not a matrix made by"):
        ctx.geno_write_rows(view, 0, geno[:1])
    ctx.geno_free(view); ctx.geno_free(g); out.free(); dA.free()
    ctx.close()


def test_pinned_staging_buffer_round_trip():
    from sfgwas_amd import capi
    import ctypes as C
    ctx = capi.Context(ol.Q_PN14, ol.P_PN14)
    p = C.c_void_p()
    ctx.check(capi.lib().sfg_pinned_alloc(ctx.h, C.byref(p), 1 << 20), "pinned_alloc")
    buf = np.ctypeslib.as_array((C.c_int8 * (1 << 20)).from_address(p.value))
    rows = np.random.default_rng(3).integers(-1, 3, (1024, 1024)).astype(np.int8)
    buf[:] = rows.reshape(-1)
    g = ctx.geno_create(1024, 1024)
    ctx.check(capi.lib().sfg_geno_write_rows(ctx.h, g, 0, 1024, p, 1024), "write_rows")
    back = np.zeros_like(rows)
    ctx.check(capi.lib().sfg_geno_download(ctx.h, g, back.ctypes.data), "download")
    assert np.array_equal(back, rows)
    ctx.geno_free(g)
    ctx.check(capi.lib().sfg_pinned_free(ctx.h, p), "pinned_free")
    ctx.close()


def test_three_ranks_scatter_chunks_to_their_windows_and_recognise_the_transpose(monkeypatch):
    """sfg_mgpu_geno_create / _write_rows / _compare_rows on 3 ranks of one device: every rank's shard holds its SNP-block window of every chunk; the rows of X^T are
    compared by the ranks whose windows hold those columns; the sharded product equals the whole-upload one (itself oracle-checked in tests/test_gpu_mgpu.py)."""
    from sfgwas_amd import capi
    from test_gpu_mgpu import make_engine, ROTS, S, LEVEL, L
    rnd = np.random.default_rng(62)
    nrow, ncol = 150, 2 * 8192 + 300                                     # 3 SNP blocks, one per rank
    geno = rnd.integers(-1, 3, (nrow, ncol)).astype(np.int8)
    mg = make_engine(monkeypatch, [0, 0, 0])
    try:
        g = mg.geno_create(nrow, ncol)
        for lo, hi in chunks(nrow, 64):                                  # 64 + 64 + 22
            mg.geno_write_rows(g, lo, geno[lo:hi])
        gw = mg.geno_upload(geno)
        lib = capi.lib()
        for i in range(mg.nlocal):                                       # shard bytes = the rank's column window
            b0, b1 = mg.geno_blocks(g, i)
            w = min(b1 * 8192, ncol) - b0 * 8192
            back = np.zeros((nrow, w), dtype=np.int8)
            mg.ctx[i].check(lib.sfg_geno_download(mg.ctx[i].h, capi.C.c_void_p(lib.sfg_mgpu_geno_shard(g, i)), back.ctypes.data), "download")
            assert np.array_equal(back, geno[:, b0 * 8192:b0 * 8192 + w]), i
        A = mg.ctx[0].fill_uniform_cts(S * 1, LEVEL, 0xC1); Ah = A.host().reshape(S, 1, 2, LEVEL + 1, mg.N).copy(); A.free()
        assert np.array_equal(mg.matmul(Ah, S, LEVEL, L, g, 0), mg.matmul(Ah, S, LEVEL, L, gw, 0))
        gt = np.ascontiguousarray(geno.T)
        assert sum(mg.geno_compare_rows(g, T, lo, gt[lo:hi]) for lo, hi in chunks(ncol, 5000)) == 0      # chunks straddle the ranks' windows
        assert sum(mg.geno_compare_rows(g, 0, lo, geno[lo:hi]) for lo, hi in chunks(nrow, 70)) == 0
        gt[8192 + 5, 7] ^= 1; gt[ncol - 1, nrow - 1] ^= 2
        assert sum(mg.geno_compare_rows(g, T, lo, gt[lo:hi]) for lo, hi in chunks(ncol, 5000)) == 2
        with pytest.raises(capi.SfgError, match="rows"):
            mg.geno_write_rows(g, nrow - 1, geno[:2])
        mg.geno_free(g); mg.geno_free(gw)
    finally:
        mg.close()
