"""SURVEY §8e as a product entry point: the multi-GPU engine sfg_mgpu_* (sfgwas_amd/csrc/mgpu.hip) gives the words of the single-GPU products
(MatMult4StreamCompute, gwas/matmult.go:1043-1236, both orientations of gwas/pca.go:344,352) for every world size and transport a one-GPU box can run:
  * one process, world 1, the exchange forced over RCCL (ncclCommInitAll + ncclReduceScatter / ncclAllReduce at one rank);
  * one process, world 2 and 3 with every rank on device 0 - RCCL refuses a repeated device, so the engine takes its in-process `direct` transport
    (a rank sums its slice out of its peers' buffers): the SAME sequence (SNP-block shards, per-column reduce-scatter over padded giant slots, reduce,
    finalize of the owned giants, all-reduce, reduce), including a rank that owns no SNP block and the unpipelined form;
  * the multi-process entry (sfg_mgpu_create_rank + a 128-byte id) at world 1.
The single-GPU products themselves are checked against the oracle elsewhere (tests/test_gpu_matmul.py, test_gpu_fullsize.py)."""
import ctypes as C
import numpy as np
import pytest

import oracle_lib as ol

pytestmark = pytest.mark.gpu
SLOTS, D, L, LEVEL, S = 8192, 91, 5, 5, 2
NROW, NCOL = 2 * SLOTS + 300, 2 * SLOTS + 77         # 3 output block columns of Q'X^T (the j >= 2 buffer hand-over), 3 SNP blocks
T, SQ = 2, 1
ROTS = list(range(1, D)) + [g * D for g in range(1, D) if g * D < SLOTS]


@pytest.fixture(scope="module")
def ref():
    """the single-GPU products of one context on the whole matrix"""
    from sfgwas_amd import capi
    ctx = capi.Context(ol.Q_PN14, ol.P_PN14)
    lib = capi.lib()
    ctx.check(lib.sfg_fill_rotkeys_synthetic(ctx.h, (C.c_int * len(ROTS))(*ROTS), len(ROTS), 0xBEEF), "keys")
    rng = np.random.default_rng(85)
    geno = rng.integers(0, 3, (NROW, NCOL), dtype=np.int8)
    geno[rng.random((NROW, NCOL)) < 0.01] = -1
    small = np.ascontiguousarray(geno[:100, :SLOTS + 5])          # 2 SNP blocks: with 3 ranks, rank 0 owns none
    nbr, mct = 3, 3
    A = {0: ctx.fill_uniform_cts(S * nbr, LEVEL, 0xB1), T: ctx.fill_uniform_cts(S * mct, LEVEL, 0xB2)}
    As = {0: ctx.fill_uniform_cts(S * 1, LEVEL, 0xB3), T: ctx.fill_uniform_cts(S * 2, LEVEL, 0xB4)}
    Ah = {f: a.host().reshape(S, -1, 2, LEVEL + 1, ctx.N) for f, a in A.items()}
    Ash = {f: a.host().reshape(S, -1, 2, LEVEL + 1, ctx.N) for f, a in As.items()}
    g = ctx.geno_upload(geno)
    gs = ctx.geno_upload(small)
    want, wants = {}, {}
    for f in (0, T, SQ, T | SQ):
        o = ctx.matmul_resident(A[f & T], S, LEVEL, L, g, f); want[f] = o.host().copy(); o.free()
    for f in (0, T):
        o = ctx.matmul_resident(As[f & T], S, LEVEL, L, gs, f); wants[f] = o.host().copy(); o.free()
    ctx.geno_free(g); ctx.geno_free(gs)
    for a in list(A.values()) + list(As.values()):
        a.free()
    ctx.close()
    return geno, small, Ah, Ash, want, wants


def make_engine(monkeypatch, devices, env=None, **kw):
    from sfgwas_amd import capi
    for k in ("SFG_MGPU_TRANSPORT", "SFG_MGPU_FORCE_COLLECTIVES", "SFG_MGPU_CACHE_GB"):
        monkeypatch.delenv(k, raising=False)
    for k, v in (env or {}).items():
        monkeypatch.setenv(k, v)
    mg = capi.MultiGpu(ol.Q_PN14, ol.P_PN14, devices=devices, **kw)
    mg.fill_rotkeys_synthetic(ROTS, 0xBEEF)
    return mg


def check_products(mg, geno, Ah, want, flag_sets):
    g = mg.geno_upload(geno)
    try:
        for f in flag_sets:
            got = mg.matmul(Ah[f & T], S, LEVEL, L, g, f)
            assert got.shape == want[f].shape
            assert np.array_equal(got, want[f]), f"flags {f}: {np.count_nonzero(got != want[f])} words differ"
    finally:
        mg.geno_free(g)


def test_world_1_exchange_over_rccl(ref, monkeypatch):
    """the exchange forced at one rank (a test switch): ncclCommInitAll, ncclReduceScatter on the collectives' queue beside the next column, ncclAllReduce"""
    geno, small, Ah, Ash, want, wants = ref
    mg = make_engine(monkeypatch, [0], {"SFG_MGPU_FORCE_COLLECTIVES": "1"})
    try:
        assert (mg.world, mg.nlocal, mg.transport) == (1, 1, "rccl")
        check_products(mg, geno, Ah, want, (0, T, T | SQ))
    finally:
        mg.close()


@pytest.mark.parametrize("n", [2, 3])
def test_direct_transport_on_one_device(ref, monkeypatch, n):
    geno, small, Ah, Ash, want, wants = ref
    mg = make_engine(monkeypatch, [0] * n)
    try:
        assert (mg.world, mg.nlocal, mg.transport) == (n, n, "direct")
        check_products(mg, geno, Ah, want, (0, T) if n == 3 else (0, T, SQ, T | SQ))
        if n == 3:                                      # 2 SNP blocks on 3 ranks: rank 0 contributes zeros to every exchange and multiplies nothing in Q X
            check_products(mg, small, Ash, wants, (0, T))
    finally:
        mg.close()


def test_unpipelined_form_and_device_pointer_entry(ref, monkeypatch):
    """SFG_MGPU_CACHE_GB=0: a rank's own rotation cache 'does not fit' -> the library's grouped cache, reduce-scatters after the product (windows that run into
    the next block column).  Driven through sfg_mgpu_matmul_dev with per-rank device buffers, as bench.py drives it."""
    from sfgwas_amd import capi
    geno, small, Ah, Ash, want, wants = ref
    mg = make_engine(monkeypatch, [0, 0], {"SFG_MGPU_CACHE_GB": "0"})
    g = mg.geno_upload(geno)
    try:
        N = mg.N
        A, out = [], []
        for i in range(mg.nlocal):
            b0, b1 = mg.geno_blocks(g, i)
            A.append(capi.DevArray.from_host(mg.ctx[i], np.ascontiguousarray(Ah[T][:, b0:b1])))
            out.append(capi.DevArray(mg.ctx[i], (S, 3, 2, L, N)))
        mg.matmul_dev(A, S, LEVEL, L, g, T, out)
        mg.sync()
        for i in range(mg.nlocal):                      # the complete result on EVERY rank
            assert np.array_equal(out[i].host(), want[T]), i
        for a in A + out:
            a.free()
        A, out = [], []
        for i in range(mg.nlocal):
            b0, b1 = mg.geno_blocks(g, i)
            A.append(capi.DevArray.from_host(mg.ctx[i], Ah[0]))
            out.append(capi.DevArray(mg.ctx[i], (S, b1 - b0, 2, L, N)))
        mg.matmul_dev(A, S, LEVEL, L, g, 0, out)
        mg.sync()
        for i in range(mg.nlocal):
            b0, b1 = mg.geno_blocks(g, i)
            assert np.array_equal(out[i].host(), want[0][:, b0:b1]), i
        for a in A + out:
            a.free()
    finally:
        mg.geno_free(g)
        mg.close()


def test_multi_process_entry_at_world_1(ref, monkeypatch):
    """sfg_mgpu_unique_id + sfg_mgpu_create_rank: how bench.py's ranks (one process per GPU) join; here one rank, the exchange forced over RCCL"""
    from sfgwas_amd import capi
    geno, small, Ah, Ash, want, wants = ref
    for k in ("SFG_MGPU_TRANSPORT", "SFG_MGPU_CACHE_GB"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("SFG_MGPU_FORCE_COLLECTIVES", "1")
    uid = capi.MultiGpu.unique_id()
    assert len(uid) == 128
    mg = capi.MultiGpu(ol.Q_PN14, ol.P_PN14, rank=0, world=1, uid=uid, device=0)
    mg.fill_rotkeys_synthetic(ROTS, 0xBEEF)
    try:
        assert (mg.world, mg.nlocal, mg.transport) == (1, 1, "rccl")
        check_products(mg, small, Ash, wants, (0, T))
    finally:
        mg.close()


def test_errors_are_reported_not_hung(ref, monkeypatch):
    from sfgwas_amd import capi
    geno, small, Ah, Ash, want, wants = ref
    with pytest.raises(capi.SfgError, match="bad device"):
        capi.MultiGpu(ol.Q_PN14, ol.P_PN14, devices=[0, 99])
    mg = make_engine(monkeypatch, [0, 0])
    g = mg.geno_upload(small)
    try:
        with pytest.raises(capi.SfgError, match="level"):          # every rank fails before the first meeting
            mg.matmul(Ash[T][:, :, :, :3], S, 2, L, g, T)
        got = mg.matmul(Ash[0], S, LEVEL, L, g, 0)                  # the engine is still usable
        assert np.array_equal(got, wants[0])
    finally:
        mg.geno_free(g)
        mg.close()


def test_association_scan_batches_round_robin_over_ranks(tmp_path, monkeypatch):
    """sfg_mgpu_assoc_stream_bed / _pgen: batch k of GenoBlockMult's loop (gwas/assoc.go:360-408) goes to rank k % world; outputs and padded column sums land where the
    single-GPU scan (tests/test_gpu_stream.py: oracle-checked) puts them.  7 batches over 2 and 3 ranks, row + column filters, SFG_SQUARE; then a .pgen."""
    from sfgwas_amd import capi
    from test_gpu_stream import write_bed
    import pgen_writer
    lib = capi.lib()
    ns, nv, batch, s, level, maxl = 130, 700, 100, 3, 5, 5
    rnd = np.random.default_rng(23)
    geno = rnd.choice(np.array([2, -1, 1, 0], dtype=np.int8), size=(ns, nv), p=[0.2, 0.05, 0.35, 0.4])
    rowf = (rnd.random(ns) < 0.9).astype(np.uint8); colf = (rnd.random(nv) < 0.9).astype(np.uint8)
    path = str(tmp_path / "chr.bed")
    write_bed(path, geno)
    N, slots = 16384, 8192
    ctx = capi.Context(ol.Q_PN14, ol.P_PN14)
    ctx.check(lib.sfg_fill_rotkeys_synthetic(ctx.h, (C.c_int * len(ROTS))(*ROTS), len(ROTS), 0xBEEF), "keys")
    dA = ctx.fill_uniform_cts(s, level, 0xA55)
    A = dA.host().copy()
    cap = 9
    want = {}
    for flags in (0, SQ):
        dout = capi.DevArray(ctx, (s, cap, 2, maxl, N))
        ctx.check(lib.sfg_memcpy_h2d(ctx.h, dout.p, np.zeros(s * cap * 2 * maxl * N, dtype=np.uint64).ctypes.data_as(C.c_void_p), s * cap * 2 * maxl * N * 8), "zero")
        sums = np.full(cap * slots, -7.0); sq = np.full(cap * slots, -7.0); n_ct = C.c_size_t()
        ctx.check(lib.sfg_assoc_stream_bed(ctx.h, path.encode(), ns, nv, rowf.ctypes.data_as(C.c_void_p), colf.ctypes.data_as(C.c_void_p), batch, dA.p, s, level, maxl, flags,
                                           dout.p, cap, C.byref(n_ct), sums.ctypes.data_as(C.c_void_p), sq.ctypes.data_as(C.c_void_p)), "assoc_stream_bed")
        assert n_ct.value == 7
        want[flags] = (dout.host().copy(), sums, sq)
        dout.free()
    # a .pgen of the same shape through the independent writer (tests/pgen_writer.py)
    codes, vrt = pgen_writer.synthetic(nv, ns, 5)                                          # every record type, LD runs that cross batch boundaries
    ppath = str(tmp_path / "chr.pgen")
    open(ppath, "wb").write(pgen_writer.write_pgen(codes, vrt, wmode=5).tobytes())
    dout = capi.DevArray(ctx, (s, cap, 2, maxl, N))
    ctx.check(lib.sfg_memcpy_h2d(ctx.h, dout.p, np.zeros(s * cap * 2 * maxl * N, dtype=np.uint64).ctypes.data_as(C.c_void_p), s * cap * 2 * maxl * N * 8), "zero")
    n_ct = C.c_size_t()
    ctx.check(lib.sfg_assoc_stream_pgen(ctx.h, ppath.encode(), rowf.ctypes.data_as(C.c_void_p), colf.ctypes.data_as(C.c_void_p), batch, dA.p, s, level, maxl, 0,
                                        dout.p, cap, C.byref(n_ct), None, None), "assoc_stream_pgen")
    want_pgen = dout.host().copy()
    dout.free(); dA.free(); ctx.close()
    kept = int(rowf.sum())
    for n in (2, 3):
        mg = make_engine(monkeypatch, [0] * n)
        try:
            for flags in (0, SQ):
                out = np.zeros((s, cap, 2, maxl, N), dtype=np.uint64)
                sums = np.full(cap * slots, -7.0); sq = np.full(cap * slots, -7.0); n_ct = C.c_size_t()
                mg.check(lib.sfg_mgpu_assoc_stream_bed(mg.h, path.encode(), ns, nv, rowf.ctypes.data_as(C.c_void_p), colf.ctypes.data_as(C.c_void_p), batch,
                                                       capi.p64(A), s, level, maxl, flags, capi.p64(out), cap, C.byref(n_ct),
                                                       sums.ctypes.data_as(C.c_void_p), sq.ctypes.data_as(C.c_void_p)), "mgpu_assoc_stream_bed")
                assert n_ct.value == 7
                assert np.array_equal(out, want[flags][0]), (n, flags)
                assert np.array_equal(sums, want[flags][1]) and np.array_equal(sq, want[flags][2]), (n, flags)
            out = np.zeros((s, cap, 2, maxl, N), dtype=np.uint64)
            mg.check(lib.sfg_mgpu_assoc_stream_pgen(mg.h, ppath.encode(), rowf.ctypes.data_as(C.c_void_p), colf.ctypes.data_as(C.c_void_p), kept, batch,
                                                    capi.p64(A), s, level, maxl, 0, capi.p64(out), cap, C.byref(n_ct), None, None), "mgpu_assoc_stream_pgen")
            assert np.array_equal(out, want_pgen), n
        finally:
            mg.close()


def test_adopted_shards_packed_residency_and_the_plaintext_cache(ref, monkeypatch):
    """the other ways a sharded matrix comes to be: per-rank windows made by the caller on the ranks' own contexts (sfg_mgpu_geno_adopt; a wrong window is refused),
    2-bit packed shards (sfg_geno_pack per rank), and the per-rank plaintext coefficient cache (filled by Q X, hit through the automorphism permutation by Q' X^T)"""
    from sfgwas_amd import capi
    geno, small, Ah, Ash, want, wants = ref
    lib = capi.lib()
    mg = make_engine(monkeypatch, [0, 0])
    try:
        shards, packed = [], []
        for i in range(mg.nlocal):
            v = [C.c_size_t() for _ in range(4)]
            assert lib.sfg_mgpu_shard(mg.world, NCOL, mg.ranks[i], *[C.byref(x) for x in v]) == 0
            c0, c1 = v[2].value, v[3].value
            g_i = mg.ctx[i].geno_upload(np.ascontiguousarray(np.where(geno[:, c0:c1] < 0, -1, geno[:, c0:c1])))
            p_i = C.c_void_p()
            mg.ctx[i].check(lib.sfg_geno_pack(mg.ctx[i].h, g_i, C.byref(p_i)), "geno_pack")
            mg.ctx[i].geno_free(g_i)
            packed.append(p_i)
        arr = (C.c_void_p * mg.nlocal)(*packed)
        g = C.c_void_p()
        bad = (C.c_void_p * mg.nlocal)(packed[1], packed[0])
        assert lib.sfg_mgpu_geno_adopt(mg.h, NROW, NCOL, bad, C.byref(g)) != 0 and b"window" in lib.sfg_mgpu_last_error(mg.h)
        mg.check(lib.sfg_mgpu_geno_adopt(mg.h, NROW, NCOL, arr, C.byref(g)), "adopt")
        mg.check(lib.sfg_mgpu_geno_set_plaintext_cache(mg.h, g, C.c_size_t(4 << 30)), "plaintext cache")
        for f in (0, T, 0):
            got = mg.matmul(Ah[f & T], S, LEVEL, L, g, f)
            assert np.array_equal(got, want[f]), f
        hits = 0
        for i in range(mg.nlocal):
            v = [C.c_size_t() for _ in range(4)]
            mg.ctx[i].check(lib.sfg_geno_plaintext_cache_stats(mg.ctx[i].h, C.c_void_p(lib.sfg_mgpu_geno_shard(g, i)), *[C.byref(x) for x in v]), "stats")
            assert v[0].value > 0
            hits += v[2].value
        assert hits > 0
        mg.geno_free(g)                                    # owns the adopted handles
    finally:
        mg.close()


def test_preflight_and_what_the_communicator_reports(ref, monkeypatch):
    """sfg_mgpu_preflight pushes a known uint64 pattern through the engine's reduce-scatter and all-reduce (the functions the products use) and checks it on the host;
    sfg_mgpu_comm_info reads the rank count off the RCCL communicator itself.  World 1 over RCCL (forced exchange), worlds 2 and 3 over the direct transport."""
    mg = make_engine(monkeypatch, [0], {"SFG_MGPU_FORCE_COLLECTIVES": "1"})
    try:
        assert mg.transport == "rccl"
        mg.preflight(4096)
        mg.preflight(1)
        assert mg.comm_info(0) == (1, 0)                 # ncclCommCount, ncclCommUserRank
        with pytest.raises(Exception, match="out of range"):
            mg.comm_info(3)
    finally:
        mg.close()
    for n in (2, 3):
        mg = make_engine(monkeypatch, [0] * n)
        try:
            mg.preflight(1000)
            assert mg.comm_info(n - 1) == (0, 0)         # the direct transport holds no communicator
        finally:
            mg.close()
    mg = make_engine(monkeypatch, [0])                   # world 1, no exchange: nothing to check, and that is not an error
    try:
        mg.preflight(16)
        assert mg.comm_info(0) == (0, 0)
    finally:
        mg.close()


def test_two_rank_engine_products_against_the_oracle_directly():
    """The engine's own oracle test at the C-ABI (not through the C++ mirror): 2 ranks over the direct transport, X = 70 x (8192 + 40) - two SNP blocks, one per rank.
    Q X (output-sharded: each rank one block column) and Q' X^T (contraction-sharded: per-column reduce-scatter over the padded giant slots, reduce, finalize of the
    owned giants, all-reduce) against orc_matmult4stream (gwas/matmult.go:1043-1236, 1238-1505) with real key-switching keys: every output word."""
    from sfgwas_amd import capi
    ring = ol.Ring(14, ol.Q_PN14, ol.P_PN14)
    keys = ol.RotKeys(ring)
    rnd = np.random.default_rng(43)
    nrow, ncol, s, level = 70, SLOTS + 40, 2, 5
    geno = rnd.integers(-1, 3, (nrow, ncol)).astype(np.int8)
    mg = capi.MultiGpu(ol.Q_PN14, ol.P_PN14, devices=[0, 0])
    try:
        assert mg.transport == "direct"
        for k in sorted(set(range(1, D)) | {g * D for g in range(1, D) if g * D < SLOTS}):
            key = capi.random_rotkey(ring.moduli, ring.beta, ring.N, 700 + k)
            keys.add(ring.galois(k), key)
            mg.load_rotkey(ring.galois(k), key)
        g = mg.geno_create(nrow, ncol)                                   # registered the way MatMult4StreamPreprocess reads it: row chunks
        for lo in range(0, nrow, 32):
            mg.geno_write_rows(g, lo, geno[lo:lo + 32])
        A = np.stack([np.stack([ring.fill_uniform(level, 150 + i)]) for i in range(s)])                          # Q : s x 1 block row
        AT = np.stack([np.stack([ring.fill_uniform(level, 170 + 2 * i + b) for b in range(2)]) for i in range(s)])  # Q': s x 2 SNP blocks
        got = mg.matmul(A, s, level, L, g, 0)
        want, _, _ = ol.matmult4stream(ring, keys, 2.0 ** 34, A, level, L, geno)
        assert np.array_equal(got, want), f"Q X: {np.count_nonzero(got != want)} words differ"
        got_t = mg.matmul(AT, s, level, L, g, T)
        want_t, _, _ = ol.matmult4stream(ring, keys, 2.0 ** 34, AT, level, L, np.ascontiguousarray(geno.T))
        assert np.array_equal(got_t, want_t), f"Q' X^T: {np.count_nonzero(got_t != want_t)} words differ"
        mg.geno_free(g)
    finally:
        mg.close()


@pytest.mark.parametrize("victim", ["mg.mine:2", "mm.skew:3"])
def test_scratch_eviction_inside_an_engine_call_keeps_the_calls_own_buffers(ref, monkeypatch, victim):
    """sfg_mgpu_matmul asks for its I/O buffers (mg.Ain, mg.Oout) in a first pass over the ranks and multiplies in a second; when the device is full the
    scratch pool gives back what EARLIER top-level calls asked for - which must not include the running call's own input and output.  SFG_TEST_SCRATCH_OOM (test switch)
    makes the n-th request of one buffer behave as 'device full' on every rank, in the SECOND product of an engine (so that buffers of an earlier call exist): the
    eviction path runs with mg.Ain / mg.Oout live, and every output word must still be right."""
    geno, small, Ah, Ash, want, wants = ref
    mg = make_engine(monkeypatch, [0, 0], {"SFG_ENABLE_TEST_HOOKS": "1", "SFG_TEST_SCRATCH_OOM": victim})
    g = mg.geno_upload(geno)
    try:
        for rep in range(2):
            got = mg.matmul(Ah[T], S, LEVEL, L, g, T)
            assert np.array_equal(got, want[T]), f"call {rep}: {np.count_nonzero(got != want[T])} words differ"
        got = mg.matmul(Ah[0], S, LEVEL, L, g, 0)
        assert np.array_equal(got, want[0])
    finally:
        mg.geno_free(g)
        mg.close()
