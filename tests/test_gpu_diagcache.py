"""sfg_matmul_from_cache: MatMult4StreamCompute on DiagCache files in the REFERENCE's byte format (filestream.go:19-282) written
by the oracle the way MatMult4StreamPreprocess writes them (matmult.go:914-1041: per block row, per active diagonal, one record
holding the Montgomery-form NTT plaintext of every block column, big-endian words).  The product from the files must equal the
on-the-fly product of the same matrix, bit for bit, and the oracle's MatMult4Stream."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
SLOTS, D, N, L, LEVEL = 8192, 91, 16384, 5, 5
SCALE = 2.0 ** 34


def write_cache(ring, logical, prefix, shifts_per_block_row=None):
    """MatMult4StreamPreprocess restated with oracle primitives.  logical: the operand (rows x cols) int8 with missing already
    cleaned by the caller's GenoFileStream semantics (negatives -> 0 here).  shifts_per_block_row[bi]: write only these diagonals
    (all others must be all-zero in every block of the row, so skipping them does not change the product)."""
    Lb = ol.lib()
    nrow, ncol = logical.shape
    nbr, m_ct = (nrow - 1) // SLOTS + 1, (ncol - 1) // SLOTS + 1
    clean = np.ascontiguousarray(np.where(logical < 0, 0, logical).astype(np.int8))
    for bi in range(nbr):
        nr = min((bi + 1) * SLOTS, nrow) - bi * SLOTS
        exists = lambda sh, nc: bool(Lb.orc_get_diag_bool(nr, nc, SLOTS, -sh))
        ncs = [min((bj + 1) * SLOTS, ncol) - bj * SLOTS for bj in range(m_ct)]
        active = [sh for sh in range(SLOTS) if any(exists(sh, nc) for nc in ncs)]
        baby = np.zeros(D, dtype=np.uint8); giant = np.zeros(D, dtype=np.uint8)
        for sh in active:                                     # tables are the union over ALL active shifts (matmult.go:962-972)
            baby[sh % D] = 1; giant[sh // D] = 1
        todo = active if shifts_per_block_row is None else [sh for sh in active if sh in shifts_per_block_row[bi]]
        dc = Lb.orc_diagcache_create(f"{prefix}_{bi}.bin".encode(), D)
        Lb.orc_diagcache_set_tables(dc, baby.ctypes.data_as(C.POINTER(C.c_uint8)), giant.ctypes.data_as(C.POINTER(C.c_uint8)))
        for sh in todo:
            pv = []
            for bj in range(m_ct):
                blk = np.ascontiguousarray(clean[bi * SLOTS:bi * SLOTS + nr, bj * SLOTS:bj * SLOTS + ncs[bj]])
                dst = np.zeros(SLOTS)
                if not Lb.orc_get_diag(ol.pd(dst), ol.pi8(blk), ncs[bj], nr, ncs[bj], SLOTS, -sh):
                    pv.append(None); continue
                pt = ring.encode_ntt(np.roll(dst, D * (sh // D)), SCALE, LEVEL + 1, prec=1)     # level-5 plaintext: 6 moduli rows
                for l in range(LEVEL + 1):
                    Lb.orc_mform_vec(ol.p64(pt[l]), N, ring.moduli[l])                        # ToMontgomeryForm, matmult.go:1026
                pv.append(pt)
            arr = (ol.u64p * m_ct)(*[ol.p64(p) if p is not None else None for p in pv])
            assert Lb.orc_diagcache_write(dc, arr, m_ct, LEVEL, SCALE, N, LEVEL + 1, sh) == 0
        Lb.orc_diagcache_close(dc)
    return nbr, m_ct


@pytest.fixture(scope="module")
def env():
    from sfgwas_amd import capi
    from sfgwas_amd.params import rotations_for_matmul
    ctx = capi.Context(ol.Q_PN14, ol.P_PN14)
    ring = ol.Ring(14, ol.Q_PN14, ol.P_PN14)
    keys = ol.RotKeys(ring)
    for k in rotations_for_matmul():
        g = ring.galois(k)
        key = capi.random_rotkey(ring.moduli, ring.beta, ring.N, 1000 + k)
        keys.add(g, key); ctx.load_rotkey(g, key)
    yield ctx, ring, keys, capi
    ctx.close()


def from_cache(ctx, capi, A, prefix, nbr, m_ct, s):
    dA = capi.DevArray.from_host(ctx, A)
    out = capi.DevArray(ctx, (s, m_ct, 2, L, N))
    hdr = np.zeros(6, dtype=np.uint64)
    ctx.check(capi.lib().sfg_diagcache_header(ctx.h, prefix.encode(), 0, capi.p64(hdr)), "header")
    assert int(hdr[0]) == m_ct and int(hdr[1]) == LEVEL and int(hdr[3]) == N and int(hdr[4]) == LEVEL + 1
    ctx.check(capi.lib().sfg_matmul_from_cache(ctx.h, dA.p, s, LEVEL, L, prefix.encode(), nbr, out.p), "from_cache")
    h = out.host(); dA.free(); out.free()
    return h


def test_small_block_cache_equals_stream_and_oracle(env, tmp_path):
    ctx, ring, keys, capi = env
    rnd = np.random.default_rng(41)
    geno = rnd.integers(-1, 3, (45, 33)).astype(np.int8)
    prefix = str(tmp_path / "cache_small")
    nbr, m_ct = write_cache(ring, geno, prefix)
    s = 2
    A = np.stack([np.stack([ring.fill_uniform(LEVEL, 60 + i)]) for i in range(s)])
    got = from_cache(ctx, capi, A, prefix, nbr, m_ct, s)
    want, _, _ = ol.matmult4stream(ring, keys, SCALE, A, LEVEL, L, geno, enc_prec=1)
    assert np.array_equal(got, want)
    stream, _, _ = ctx.matmul_stream(A, LEVEL, L, geno)
    assert np.array_equal(got, stream)


def test_two_block_rows_two_block_columns_sparse_diagonals(env, tmp_path):
    """(8192 + 30) x (8192 + 17) operand: accumulation across block-row files and a record layout with two plaintexts per
    record, one of them sometimes empty.  Only 40 diagonals per block row carry non-zero genotypes and only those are written
    (a zero plaintext contributes nothing), which keeps the files at tens of MB instead of 13 GB."""
    ctx, ring, keys, capi = env
    rnd = np.random.default_rng(42)
    nrow, ncol = SLOTS + 30, SLOTS + 17
    geno = np.zeros((nrow, ncol), dtype=np.int8)
    chosen = {}
    for bi in range(2):
        nr = min((bi + 1) * SLOTS, nrow) - bi * SLOTS
        sh = sorted(set(int(x) for x in rnd.integers(0, SLOTS, 40)) | {0, 1, SLOTS - 1, 90, 91, 8190})
        chosen[bi] = set(sh)
        for bj in range(2):
            nc = min((bj + 1) * SLOTS, ncol) - bj * SLOTS
            for shv in sh:                                        # X[(j + shift) mod n][j] on diagonal `shift` of block (bi, bj)
                j = np.arange(nc); i = (j + shv) % SLOTS
                ok = i < nr
                geno[bi * SLOTS + i[ok], bj * SLOTS + j[ok]] = rnd.integers(-1, 3, int(ok.sum())).astype(np.int8)
    prefix = str(tmp_path / "cache_2x2")
    nbr, m_ct = write_cache(ring, geno, prefix, shifts_per_block_row=chosen)
    assert (nbr, m_ct) == (2, 2)
    s = 1
    A = np.stack([np.stack([ring.fill_uniform(LEVEL, 80 + b) for b in range(nbr)])])
    got = from_cache(ctx, capi, A, prefix, nbr, m_ct, s)
    g = ctx.geno_upload(geno)
    dA = capi.DevArray.from_host(ctx, A)
    fly = ctx.matmul_resident(dA, s, LEVEL, L, g)
    assert np.array_equal(got, fly.host())
    dA.free(); fly.free(); ctx.geno_free(g)
    # a missing block-row file fails loudly, like os.Open in the reference
    os.remove(prefix + "_1.bin")
    with pytest.raises(capi.SfgError):
        from_cache(ctx, capi, A, prefix, nbr, m_ct, s)


def test_gpu_written_cache_equals_the_oracle_writer_byte_for_byte_and_is_consumed(env, tmp_path):
    """sfg_diagcache_write (MatMult4StreamPreprocess, matmult.go:914-1041): the file the GPU writes from device-encoded diagonals (MForm, big-endian
    payload, filestream.go:144-231 layout) equals the oracle writer's bytes, and sfg_matmul_from_cache multiplies from it; the transposed view of
    the one resident copy gives the file of X^T; an existing file is kept."""
    ctx, ring, keys, capi = env
    lib = capi.lib()
    rnd = np.random.default_rng(43)
    geno = rnd.integers(-1, 3, (45, 33)).astype(np.int8)
    g = ctx.geno_upload(geno)
    for flags, logical, tag in ((0, geno, "x"), (capi.SFG_TRANSPOSE, np.ascontiguousarray(geno.T), "xt")):
        ref_prefix, gpu_prefix = str(tmp_path / f"ref_{tag}"), str(tmp_path / f"gpu_{tag}")
        nbr, m_ct = write_cache(ring, logical, ref_prefix)
        nfiles = C.c_int()
        ctx.check(lib.sfg_diagcache_write(ctx.h, g, flags, LEVEL, gpu_prefix.encode(), C.byref(nfiles)), "diagcache_write")
        assert nfiles.value == nbr == 1
        a, b = open(ref_prefix + "_0.bin", "rb").read(), open(gpu_prefix + "_0.bin", "rb").read()
        assert len(a) == len(b) and a == b, "GPU-written DiagCache differs from the oracle writer's bytes"
        s = 2
        A = np.stack([np.stack([ring.fill_uniform(LEVEL, 90 + i)]) for i in range(s)])
        got = from_cache(ctx, capi, A, gpu_prefix, nbr, m_ct, s)
        want, _, _ = ol.matmult4stream(ring, keys, SCALE, A, LEVEL, L, logical, enc_prec=1)
        assert np.array_equal(got, want)
        ctx.check(lib.sfg_diagcache_write(ctx.h, g, flags, LEVEL, gpu_prefix.encode(), C.byref(nfiles)), "diagcache_write again")
        assert nfiles.value == 0                                          # "skips existing files" (matmult.go:928-931)
    ctx.geno_free(g)
