#!/usr/bin/env python3
"""bench.py — one PCA power iteration's local work: the two hot products
   Q*X   (kp x n_ind)(n_ind x m_snp)   pca.go:344 -> MatMult4StreamCompute
   Q'*X^T (kp x m_snp)(m_snp x n_ind)  pca.go:352 -> MatMult4StreamCompute
on synthetic data, through the C-ABI of libsfgwas_hip.so, one process per GPU.

    python bench.py --gpus N --steps K --warmup W [--config c4|c3|c2|tiny]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Sharding (strong scaling, total work fixed): the genotype matrix is split by SNP block (8192 columns of X) across
ranks.  Q*X is output-sharded (no collective).  Q'*X^T is contraction-sharded: ranks all-reduce the uint64
accumulators over RCCL BEFORE the giant-step rotations (key switching is not bit-linear), each rank then aligns its
share of the giant steps and the aligned outputs are all-reduced again (256 MB at 100k x 1M).

Prints ONE JSON line (rank 0).  PyTorch is plumbing here: device tensors, streams and torch.distributed.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

CONFIGS = {          # n_ind, m_snp  
    "c4": (100_000, 1_000_000),
    "c3": (50_000, 500_000),
    "c2": (10_000, 100_000),
    "tiny": (8_192, 24_576),
}
SLOTS, D, N, L, LEVEL, KP = 8192, 91, 16384, 5, 5, 15
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


def ceil_div(a, b):
    return (a + b - 1) // b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default=os.environ.get("SFG_BENCH_CONFIG", "c4"))
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="bounded CPU-baseline sample (seconds of wall time)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from sfgwas_amd import capi
    import oracle_lib as ol

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # SFG_BENCH_FORCE_COLLECTIVES=1 runs the RCCL collectives even at world size 1 (used to exercise the N > 1 code path
    # on a single-GPU box under torch.distributed.run)
    force_coll = os.environ.get("SFG_BENCH_FORCE_COLLECTIVES") == "1"
    use_dist = world > 1 or (force_coll and "RANK" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    n_ind, m_snp = CONFIGS[args.config]
    nbr_x, mct_x = ceil_div(n_ind, SLOTS), ceil_div(m_snp, SLOTS)          # block rows / cols of X
    # SNP-block shard of this rank
    from sfgwas_amd.sharding import snp_block_range, giant_range
    blk0, blk1, c0, c1 = snp_block_range(m_snp, rank, world)
    m_loc, nblk_loc = c1 - c0, blk1 - blk0

    ctx = capi.Context(ol.Q_PN14, ol.P_PN14, device=local_rank)
    lib = capi.lib()
    stream = torch.cuda.current_stream()
    lib.sfg_ctx_set_stream(ctx.h, C.c_void_p(stream.cuda_stream))

    def chk(rc, what):
        ctx.check(rc, what)

    # ---- synthetic inputs, generated on the device (seeds: BASELINE.md §2)
    geno = torch.empty((n_ind, m_loc), dtype=torch.int8, device=dev)
    chk(lib.sfg_fill_geno_dev(ctx.h, C.c_void_p(geno.data_ptr()), n_ind, m_loc, 0x5F6A + 131 * rank), "fill_geno")
    gh = C.c_void_p()
    chk(lib.sfg_geno_from_device(ctx.h, C.c_void_p(geno.data_ptr()), n_ind, m_loc, m_loc, C.byref(gh)), "geno_from_device")
    ctw = 2 * (LEVEL + 1) * N
    A1 = torch.empty((KP, nbr_x, ctw), dtype=torch.int64, device=dev)       # Q   : kp x n_ind
    A2 = torch.empty((KP, nblk_loc, ctw), dtype=torch.int64, device=dev)    # Q'  : kp x m_snp, this rank's SNP blocks
    chk(lib.sfg_fill_uniform_ct_dev(ctx.h, C.c_void_p(A1.data_ptr()), KP * nbr_x, LEVEL, 0xC1F3), "fill A1")
    chk(lib.sfg_fill_uniform_ct_dev(ctx.h, C.c_void_p(A2.data_ptr()), KP * nblk_loc, LEVEL, 0xC1F3 + 7919 * (rank + 1)), "fill A2")
    rots = list(range(1, D)) + [g * D for g in range(1, D) if g * D < SLOTS]
    arr = (C.c_int * len(rots))(*rots)
    chk(lib.sfg_fill_rotkeys_synthetic(ctx.h, arr, len(rots), 0xBEEF), "fill rotkeys")
    outw = 2 * L * N
    out1 = torch.empty((KP, nblk_loc, outw), dtype=torch.int64, device=dev)
    out2 = torch.empty((KP, nbr_x, outw), dtype=torch.int64, device=dev)
    acc2 = torch.empty((nbr_x, D, KP, outw), dtype=torch.int64, device=dev)
    ctx.sync()

    phase_tot = {}

    def add_phases():
        for ph in ("rotate", "skew", "encode", "mac", "mac_small", "mac_big"):
            ms = ctx.phase_ms(ph)
            if ms >= 0:
                n = lib.sfg_last_phase_launches(ctx.h, ph.encode())
                a = phase_tot.setdefault(ph, [0.0, 0, 0.0])
                a[0] += ms
                a[1] += n
                a[2] += max(lib.sfg_last_phase_bytes(ctx.h, ph.encode()), 0.0)

    def step():
        # (1) Q * X : output block columns of this rank
        chk(lib.sfg_matmul_resident_dev(ctx.h, C.c_void_p(A1.data_ptr()), KP, LEVEL, L, gh, 0, C.c_void_p(out1.data_ptr())), "Q*X")
        add_phases()
        # (2) Q' * X^T : contraction over this rank's SNP blocks, combined before the giant steps
        lib.sfg_ctx_clear_phases(ctx.h)
        chk(lib.sfg_matmul_accumulate_dev(ctx.h, C.c_void_p(A2.data_ptr()), KP, LEVEL, L, gh, capi.SFG_TRANSPOSE,
                                          0, nblk_loc, 0, nbr_x, 0, C.c_void_p(acc2.data_ptr())), "Q'*X^T accumulate")
        if use_dist:
            for j in range(nbr_x):                                          # one collective per output block column: 1.8 GB each,
                dist.all_reduce(acc2[j])                                    # element counts stay below 2^31; sums < 8 * 2^46, no overflow
            chk(lib.sfg_reduce_rows_dev(ctx.h, C.c_void_p(acc2.data_ptr()), nbr_x * D * KP * 2, L), "reduce acc")
        g0, g1 = giant_range(rank, world)
        chk(lib.sfg_matmul_finalize_dev(ctx.h, C.c_void_p(acc2.data_ptr()), KP, L, nbr_x, g0, g1, 0, C.c_void_p(out2.data_ptr())), "finalize")
        if use_dist:
            dist.all_reduce(out2)
            chk(lib.sfg_reduce_rows_dev(ctx.h, C.c_void_p(out2.data_ptr()), KP * nbr_x * 2, L), "reduce out")
        add_phases()

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    phase_tot.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- work accounting: useful ring-MACs = nrow*ncol*s*2 polys*L*(N/slots) per product
    macs_per_product = n_ind * m_snp * KP * 2 * L * (N // SLOTS)
    macs_per_step = 2 * macs_per_product
    value = macs_per_step * args.steps / dt
    res = {
        "metric": "pca_power_iter_ring_macs_per_s", "value": value, "unit": "ring-MAC/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64 (exact integers in fp64 limbs)",
        "data": "synthetic",
        "config": {"workload": f"{args.config}: one PCA power iteration local work = Q*X + Q'*X^T, {n_ind} x {m_snp} int8 genotypes, "
                               f"kp={KP}, PN14QP438-shaped ring (N=16384, L=5 of 6 moduli), on-the-fly diagonal encode",
                   "parallelism": f"snp-block x{world}", "power_iter_wall_s": dt / args.steps},
    }
    if rank == 0:
        # ---- roofline of the dominant kernel (k_mac, small-modulus instance): algorithmic bytes per launch / avg duration
        ms_small, n_small, by_small = phase_tot.get("mac_small", [0.0, 0, 0.0])
        if n_small:
            # dominant kernel = k_mac_dma<false> (the four 35/36-bit moduli).  Algorithmic bytes are reported by the library
            # per launch (fp64 rot operand + half-row plaintexts + accumulators); a launch covers up to 8 block
            # rows x 1 block column, so bytes and duration are averaged over the launches of the timed region.
            avg_ms = ms_small / n_small
            per_launch = by_small / n_small
            achieved = by_small / (ms_small * 1e-3) / 1e9
            traffic = None
            try:        # HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command
                pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic_per_launch.json")))
                if pm.get("_config") == args.config:
                    ks = [k for k in pm if k.startswith("void k_mac_dma<false")]          # the small-modulus instance the default build launches
                    best = max(ks, key=lambda k: pm[k]["launches"])
                    traffic = pm[best]["hbm_bytes_per_launch"]
            except Exception:
                pass
            nl_small = L - 1
            res["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                               "kernel": "k_mac_dma<false, 1>", "avg_launch_ms": avg_ms, "launches": n_small,
                               "alg_bytes_per_launch": per_launch,
                               "padded_ring_macs_per_s_in_kernel": 2 * ceil_div(n_ind, SLOTS) * ceil_div(m_snp, SLOTS) * D * D * 2 * KP * nl_small * N
                                                                   * args.steps / (ms_small * 1e-3)}
        res["phases_ms_per_step"] = {k: v[0] / args.steps for k, v in phase_tot.items()}
        if not args.no_cpu_baseline:
            cores = os.cpu_count() or 1
            n_done = C.c_longlong()
            rate = ol.lib().orc_bench_mac(KP, L, N, cores, float(args.cpu_seconds), C.byref(n_done))
            res["cpu_baseline"] = {"value": rate, "unit": "ring-MAC/s", "cores": cores, "kind": "port",
                                   "sample": f"reference MAC loop (CPMultAccWithoutMRedV2, s={KP}, L={L}, N={N}) restated in C, "
                                             f"{cores} threads x {args.cpu_seconds:.0f} s = {n_done.value:.3e} MACs; cached-diagonal mode "
                                             f"(encode excluded); CPU restatement, not the Go binary"}
        print(json.dumps(res), flush=True)
    lib.sfg_geno_free(ctx.h, gh)
    ctx.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
