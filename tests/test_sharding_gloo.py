"""World-size-2 (gloo, CPU) check of the multi-GPU contract bench.py implements over RCCL:
  * SNP-block ranges partition the matrix;
  * Q*X     : concatenating the ranks' output block columns gives the unsharded product;
  * Q'*X^T  : summing the ranks' canonical accumulators (reduce-scatter over the padded giant axis, then mod q) BEFORE the
              giant-step rotations, then letting each rank align its giant steps and all-reducing the aligned partial
              outputs, gives the unsharded product bit for bit — whereas summing rotated partial outputs does not (key
              switching is not bit-linear).  The window arithmetic is bench.py's; gloo has no reduce-scatter, so the
              collective is emulated by all-reduce + slice (same sums).
Runs the oracle at a small ring (N = 32) so that 2 processes finish in seconds."""
import os
import sys
import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from sfgwas_amd.sharding import snp_block_range, giant_range, giant_slots, ceil_div  # noqa: E402


def test_snp_block_ranges_partition():
    for m_snp in [1, 8192, 8193, 100_000, 1_000_000]:
        for world in [1, 2, 3, 8]:
            cols = []
            for r in range(world):
                b0, b1, c0, c1 = snp_block_range(m_snp, r, world)
                assert 0 <= b0 <= b1 and c0 == b0 * 8192
                cols += list(range(c0, c1))[:1] + list(range(c0, c1))[-1:]
            last = snp_block_range(m_snp, world - 1, world)
            assert last[1] == ceil_div(m_snp, 8192) and last[3] == m_snp
    for world in [1, 2, 4, 8]:
        g = [giant_range(r, world) for r in range(world)]
        assert g[0][0] == 0 and g[-1][1] == 91 and all(g[i][1] == g[i + 1][0] for i in range(world - 1))
    for world in [1, 2, 3, 4, 8]:
        sl = [giant_slots(r, world) for r in range(world)]
        assert all(x[0] == sl[0][0] for x in sl) and sl[0][0] * world >= 91
        owned = [gi for (gpr, lo, hi) in sl for gi in range(lo, hi)]
        assert owned == list(range(91))


def _worker(rank, world, port, q):
    import oracle_lib as ol
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    logN = 5
    qs = ol.small_primes(logN, 46, 1) + ol.small_primes(logN, 35, 5)
    ps = ol.small_primes(logN, 43, 2)
    ring = ol.Ring(logN, qs, ps)
    slots, d = ring.slots, 4
    sk = ring.gen_secret(3)
    keys = ol.RotKeys(ring)
    keys.gen_for_rotations(sk, list(range(1, d)) + [g * d for g in range(1, d)])
    rnd = np.random.default_rng(7)                      # same data on both ranks
    n_ind, m_snp, s, level, L = 40, 70, 2, 5, 5
    X = rnd.integers(-1, 3, (n_ind, m_snp)).astype(np.int8)
    nbr_x, mct_x = ceil_div(n_ind, slots), ceil_div(m_snp, slots)
    A1 = np.stack([np.stack([ring.fill_uniform(level, 10 + i * 7 + b) for b in range(nbr_x)]) for i in range(s)])
    A2 = np.stack([np.stack([ring.fill_uniform(level, 90 + i * 7 + b) for b in range(mct_x)]) for i in range(s)])
    b0, b1, c0, c1 = snp_block_range(m_snp, rank, world, slots)
    Xloc = np.ascontiguousarray(X[:, c0:c1])
    # ---- Q * X: output-sharded
    full1, _, _ = ol.matmult4stream(ring, keys, 2.0 ** 34, A1, level, L, X)
    loc1, _, _ = ol.matmult4stream(ring, keys, 2.0 ** 34, A1, level, L, Xloc)
    gathered = [torch.zeros((s, mct_x, 2, L, ring.N), dtype=torch.int64) for _ in range(world)]
    pad = np.zeros((s, mct_x, 2, L, ring.N), dtype=np.uint64)
    pad[:, :loc1.shape[1]] = loc1
    dist.all_gather(gathered, torch.from_numpy(pad.view(np.int64)))
    cat = np.concatenate([g.numpy().view(np.uint64)[:, :snp_block_range(m_snp, r, world, slots)[1] - snp_block_range(m_snp, r, world, slots)[0]]
                          for r, g in enumerate(gathered)], axis=1)
    ok1 = np.array_equal(cat, full1)
    # ---- Q' * X^T: contraction-sharded
    XT = np.ascontiguousarray(X.T)
    full2, _, _ = ol.matmult4stream(ring, keys, 2.0 ** 34, A2, level, L, XT)
    m_ct = nbr_x
    acc = np.zeros((m_ct, d, s, 2, L, ring.N), dtype=np.uint64)
    rc = ol.lib().orc_matmult_accumulate(ring.h, keys.h, 2.0 ** 34, ol.p64(np.ascontiguousarray(A2)), s, level, L, ol.pi8(XT), m_snp, n_ind,
                                         0, 0, b0, b1, ol.p64(acc), None)
    assert rc == 0
    # the wrong way first: rotate the partial accumulators locally, then sum
    wrong = np.zeros((s, m_ct, 2, L, ring.N), dtype=np.uint64)
    ol.lib().orc_matmult_finalize(ring.h, keys.h, L, s, m_ct, ol.p64(acc), None, 0, d, 0, ol.p64(wrong))
    tw = torch.from_numpy(wrong.view(np.int64).copy())
    dist.all_reduce(tw)
    wrong_sum = tw.numpy().view(np.uint64)
    for l in range(L):
        wrong_sum[:, :, :, l, :] %= np.uint64(qs[l])
    # the bit-exact way (bench.py's scheme): per output block column, reduce-scatter the accumulators over the giant axis padded
    # to world * gpr slots (the window of the last real column runs into zero padding, the others into the next column: those
    # slots are ignored), reduce mod q, align this rank's giants, all-reduce the aligned partial outputs
    gpr, g_lo, g_hi = giant_slots(rank, world, d)
    kw = s * 2 * L * ring.N                               # words per giant slot of one block column
    col = d * kw
    flat = np.zeros(m_ct * col + (world * gpr - d) * kw, dtype=np.uint64)
    flat[:m_ct * col] = acc.reshape(-1)
    mine = np.zeros((m_ct, d, s, 2, L, ring.N), dtype=np.uint64)     # this rank's slots, placed at their giant index
    for j in range(m_ct):
        win = torch.from_numpy(flat[j * col: j * col + world * gpr * kw].view(np.int64).copy())
        dist.all_reduce(win)                                # gloo stand-in for reduce_scatter_tensor: same sums, then take my chunk
        chunk = win.numpy().view(np.uint64)[rank * gpr * kw:(rank + 1) * gpr * kw].reshape(gpr, s, 2, L, ring.N)
        for gslot in range(gpr):
            if g_lo + gslot < d:
                mine[j, g_lo + gslot] = chunk[gslot]
    for l in range(L):
        mine[:, :, :, :, l, :] %= np.uint64(qs[l])
    part = np.zeros((s, m_ct, 2, L, ring.N), dtype=np.uint64)
    ol.lib().orc_matmult_finalize(ring.h, keys.h, L, s, m_ct, ol.p64(mine), None, g_lo, g_hi, 0, ol.p64(part))
    tp = torch.from_numpy(part.view(np.int64))
    dist.all_reduce(tp)
    for l in range(L):
        part[:, :, :, l, :] %= np.uint64(qs[l])
    ok2 = np.array_equal(part, full2)
    differs = not np.array_equal(wrong_sum, full2)
    q.put((rank, ok1, ok2, differs))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])          # 3: the giant axis (d = 4 at this ring size) needs padding
def test_multi_rank_sharding_is_bit_exact(world):
    import oracle_lib as ol
    ol.build_oracle()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok1, ok2, differs in res:
        assert ok1, f"rank {rank}: output-sharded Q*X differs from the unsharded product"
        assert ok2, f"rank {rank}: accumulate -> all-reduce -> finalize differs from the unsharded product"
        assert differs, "summing rotated partial outputs unexpectedly matched: the two-phase API would be unnecessary"
