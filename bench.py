#!/usr/bin/env python3
"""bench.py — one PCA power iteration's local work: the two hot products
   Q*X   (kp x n_ind)(n_ind x m_snp)   pca.go:344 -> MatMult4StreamCompute
   Q'*X^T (kp x m_snp)(m_snp x n_ind)  pca.go:352 -> MatMult4StreamCompute
on synthetic data, through the C-ABI of libsfgwas_hip.so, one process per GPU.

    python bench.py --gpus N --steps K --warmup W [--config c4|c3|c2|tiny]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Sharding (strong scaling, total work fixed): the genotype matrix is split by SNP block (8192 columns of X) across
ranks; every rank generates exactly the window of the SAME global matrix it owns, so any world size multiplies the same
matrix and the output digests in the JSON line are comparable between N = 1 and N > 1.
  Q*X    : output-sharded, no data-path collective: every rank builds the baby-step rotation cache of all inputs (72 ms of key
           switching at 100k x 1M).  SFG_BENCH_ROTCACHE=sharded builds it in shards instead (each rank key-switches 1/world of the
           (block row, input) jobs, one all-gather, a scatter into the MAC layout): identical digests, 25 GB into every rank.
  Q'*X^T : contraction-sharded.  Key switching is not bit-linear, so partial sums are combined BEFORE the giant-step
           rotations: reduce-scatter of the uint64 accumulators over the giant axis, one output block column at a time while the
           next column is being multiplied; each rank aligns its giant steps, and the aligned partial outputs (256 MB at
           100k x 1M) are all-reduced.
  --backend nccl (default: RCCL over xGMI, one GPU per rank) | gloo (collectives staged through host memory: a rehearsal of the
  N > 1 code path in which several ranks may share ONE GPU; its timings mean nothing).

Before timing, rank 0 pushes a reduced problem (1 block row x 2 block columns, all 8192 diagonals, s = 2) through the
CPU oracle and compares every output word with the HIP path ("parity_gate" in the JSON line; --no-check skips it).

Prints ONE JSON line (rank 0).  PyTorch is plumbing here: device tensors, streams and torch.distributed.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {          # n_ind, m_snp  
    "c4": (100_000, 1_000_000),
    "c3": (50_000, 500_000),
    "c2": (10_000, 100_000),
    "tiny": (8_192, 24_576),
}
KP = 15                    # num_pcs_to_remove + num_oversampling (pca.go:87)
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
FP64_VALU_SPEC_FMA_S = 256 * 4 * 16 * 2.4e9          # 3.93e13: 256 CUs x 4 SIMDs x 16 fp64 lanes/clk x 2.4 GHz (78.6 TFLOP/s spec)
UBENCH_FILE = "profiles/r02_ubench_dpp_fma_rate.txt"           # committed microbenchmark (tools/ubench_dpp.hip): the MAC's own LDS-fed DPP tile, no DMA, no barrier
UBENCH_FMA_S = 3.398e13


def ceil_div(a, b):
    return (a + b - 1) // b


def oracle_lib():
    """the CPU oracle: checker (parity gate) and timed CPU baseline only"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as ol
    return ol


def parity_gate(ctx, capi, P):
    """SURVEY §8d 'Parity check during measurement': 8192 x 16384 genotypes (1 block row x 2 block columns, every one of the
    8192 diagonals of both blocks), s = 2, on the bench context's own (synthetic) keys; every output word vs the oracle."""
    import numpy as np
    ol = oracle_lib()
    t0 = time.perf_counter()
    ring = ol.Ring(P.LOGN, P.Q_PN14, P.P_PN14)
    keys = ol.RotKeys(ring)
    for k in P.rotations_for_matmul():
        g = ring.galois(k)
        keys.add(g, ctx.export_rotkey(g))
    s, nrow, ncol = 2, P.SLOTS, 2 * P.SLOTS
    gd, gh = ctx.fill_geno(nrow, ncol, 0x6A7E)
    geno = gd.host()
    A = ctx.fill_uniform_cts(s, P.MAX_LEVEL, 0x6A7F)
    out = ctx.matmul_resident(A, s, P.MAX_LEVEL, P.MAX_LEVEL, gh)
    got = out.host()
    Ah = A.host().reshape(s, 1, 2, P.MAX_LEVEL + 1, P.N)
    bad = 0
    for j in range(2):
        sub = np.ascontiguousarray(geno[:, j * P.SLOTS:(j + 1) * P.SLOTS])
        want, _, _ = ol.matmult4stream(ring, keys, P.DEFAULT_SCALE, Ah, P.MAX_LEVEL, P.MAX_LEVEL, sub, enc_prec=1)
        bad += int(np.count_nonzero(got[:, j] != want[:, 0]))
    for d in (gd, A, out):
        d.free()
    ctx.geno_free(gh)
    return {"status": "ok" if bad == 0 else "FAILED", "mismatching_words": bad, "words": int(got.size),
            "problem": f"{nrow} x {ncol} int8 (1 block row x 2 block columns, all 8192 diagonals), s={s}, vs oracle/sfgwas_oracle.c",
            "seconds": round(time.perf_counter() - t0, 1)}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def granted_cores():
    """cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a 1-GPU lease of a 256-thread host is a fraction of it)"""
    present = os.cpu_count() or 1
    try:
        aff = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        aff = present
    quota = None
    try:                                                    # cgroup v2
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:                                                # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    granted = aff if quota is None else max(1, min(aff, int(quota + 0.5)))
    return granted, {"present": present, "affinity": aff, "cgroup_quota_cpus": quota, "granted": granted}


def cpu_baseline(seconds, L, N, D):
    """the reference's MAC loop (matmult.go:247-289,380-399) restated in C, built -O3 -march=native ON this host, with the
    reference's data layout (shared rotCache[i][baby], u128 accCache[i][giant]); one thread per GRANTED core (affinity mask capped by the
    cgroup quota), every buffer first-touched before the clock starts, >= 3 whole passes over the sample; encode excluded"""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "native"], stdout=subprocess.DEVNULL)
    lib = C.CDLL(os.path.join(ROOT, "oracle", "_build", "liboracle_native.so"))
    lib.orc_bench_mac_ref_layout.restype = C.c_double
    lib.orc_bench_mac_ref_layout.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_longlong), C.POINTER(C.c_int), C.POINTER(C.c_double)]
    threads, cores = granted_cores()
    n_done, active, secs = C.c_longlong(), C.c_int(), C.c_double()
    # calibration: one pass over 2 items per thread (not reported), then the sample sized so that PASSES whole passes take about `seconds`
    passes, total_items = 3, KP * D
    cal_items = min(total_items, 2 * threads)
    cal = lib.orc_bench_mac_ref_layout(KP, L, N, D, threads, cal_items, 1, C.byref(n_done), C.byref(active), C.byref(secs))
    macs_per_item = D * 2 * L * N
    items = int(max(threads, min(total_items, cal * seconds / passes / macs_per_item)))
    passes = int(max(passes, min(64, round(cal * seconds / (items * macs_per_item)))))      # a fast host finishes three passes over ALL items early: more whole passes, same sample
    rate = lib.orc_bench_mac_ref_layout(KP, L, N, D, threads, items, passes, C.byref(n_done), C.byref(active), C.byref(secs))
    return {"value": rate, "unit": "ring-MAC/s", "cores": active.value, "cores_granted_vs_present": cores, "kind": "port", "cpu": cpu_model(),
            "build": "gcc -O3 -march=native -fopenmp (oracle/Makefile: native)", "per_core": rate / max(active.value, 1),
            "sample": f"reference MAC loop CPMultAccWithoutMRedV2 (u128 lazy accumulation, s={KP}, L={L}, N={N}) in the reference's layout: shared "
                      f"rotCache[i][baby] ({KP * D} cts), {items} of the {total_items} accCache[i][giant] of one block column (the reference's lock units) as work items, "
                      f"one plaintext per diagonal; {passes} whole passes = {n_done.value:.3e} MACs in {secs.value:.1f} s on {active.value} threads "
                      f"(cores granted {cores['granted']} of {cores['present']} present); all buffers first-touched before the clock; cached-diagonal "
                      f"mode (encode excluded); CPU restatement, not the Go binary"}


def cpu_end_to_end(ctx, P, threads, cores):
    """north_star: "the reference Go CPU path timed on the same host".  The Go binary cannot be built (no toolchain, un-vendored modules), so this is the repository's
    CPU restatement of the WHOLE MatMult4Stream (oracle/sfgwas_oracle.c: orc_matmult4stream = GetDiag, rotation-before-encode, big-float inverse embedding + NTT +
    MForm per diagonal, the baby-step key switches, the lazy u128 MAC, REDC, the giant-step key switches; gwas/matmult.go:1238-1505) on ONE 8192 x 8192 block at
    s = kp, work-shared over the giant steps with OpenMP on the granted cores, built -O3 -march=native on this host.  The restatement is a checker, not a tuned CPU
    implementation; the figure is a reported baseline, labelled as such.  Encoder precision: the double-double form (enc_prec = 1, what the parity gate runs) is timed
    on the whole block; the reference's big-float encoder is restated at 113 bits (__float128), whose extra cost per diagonal is measured on a 1024 x 1024 block
    (2047 diagonals) and added for all 8192."""
    import numpy as np
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "native"], stdout=subprocess.DEVNULL)
    os.environ["SFG_ORACLE_SO"] = os.path.join(ROOT, "oracle", "_build", "liboracle_native.so")
    ol = oracle_lib()
    ol._lib = None                                          # (the gate ran on the -O2 test build: reload as the native one)
    try:
        C.CDLL("libgomp.so.1").omp_set_num_threads(int(threads))
    except OSError:
        pass
    ring = ol.Ring(P.LOGN, P.Q_PN14, P.P_PN14)
    keys = ol.RotKeys(ring)
    for k in P.rotations_for_matmul():                      # the bench context's synthetic keys (timing-equivalent to real ones)
        g = ring.galois(k)
        keys.add(g, ctx.export_rotkey(g))
    rnd = np.random.default_rng(0xE2E)

    def run(n, s, prec):
        geno = rnd.integers(0, 3, (n, n)).astype(np.int8)
        A = np.stack([np.stack([ring.fill_uniform(P.MAX_LEVEL, 900 + i)]) for i in range(s)])
        t0 = time.perf_counter()
        ol.matmult4stream(ring, keys, P.DEFAULT_SCALE, A, P.MAX_LEVEL, P.MAX_LEVEL, geno, enc_prec=prec)
        return time.perf_counter() - t0
    t_block = run(P.SLOTS, KP, 1)
    t_dd, t_q = run(1024, 1, 1), run(1024, 1, 0)
    extra = max(t_q - t_dd, 0.0) / 2047 * P.SLOTS
    macs = P.SLOTS * P.SLOTS * KP * 2 * P.MAX_LEVEL * (P.N // P.SLOTS)
    return {"value": macs / t_block, "unit": "ring-MAC/s (useful, end to end)", "cores": int(threads), "cores_granted_vs_present": cores, "kind": "port",
            "seconds_per_block": t_block, "value_113bit_encoder": macs / (t_block + extra), "seconds_per_block_113bit_encoder": t_block + extra,
            "sample": f"orc_matmult4stream on one {P.SLOTS} x {P.SLOTS} block, s = {KP}, all 8192 diagonals: encode (double-double) + NTT + MForm + {KP * 90} baby-step and "
                      f"giant-step key switches + lazy u128 MAC + REDC, OpenMP over giant steps on {int(threads)} threads: {t_block:.1f} s; 113-bit encoder: + "
                      f"{extra:.1f} s per block (from {t_dd:.1f} s vs {t_q:.1f} s on a 1024 x 1024 block, s = 1); CPU restatement (the repository's checker), not the Go binary"}


class Coll:
    """The collectives of the N > 1 path on device tensors.  nccl: RCCL, stream-ordered by torch.distributed.  gloo: staged through host
    memory (several ranks may then share one GPU: a correctness rehearsal of the same sequence, not a measurement)."""

    class _Done:
        def wait(self):
            pass

    def __init__(self, dist, torch, backend, solo=None):
        self.dist, self.torch, self.host, self.solo = dist, torch, backend == "gloo", solo

    def reduce_scatter(self, out, inp, async_op=False):
        if self.solo:                                       # timing-only emulation of ONE rank of a larger world: its own slice, no peers
            out.copy_(inp.view(self.solo[1], -1)[self.solo[0]])
            return self._Done()
        if not self.host:
            w = self.dist.reduce_scatter_tensor(out, inp, async_op=async_op)
            return w if async_op else self._Done()
        h = inp.cpu()                                       # synchronises with the current stream
        o = self.torch.empty(out.shape, dtype=out.dtype)
        self.dist.reduce_scatter_tensor(o, h)
        out.copy_(o)
        return self._Done()

    def all_gather(self, out, inp):
        if self.solo:
            out.view(self.solo[1], -1)[self.solo[0]].copy_(inp)
            return
        if not self.host:
            self.dist.all_gather_into_tensor(out, inp)
            return
        o = self.torch.empty(out.shape, dtype=out.dtype)
        self.dist.all_gather_into_tensor(o, inp.cpu())
        out.copy_(o)

    def all_reduce(self, t, op=None):
        if self.solo:
            return
        kw = {} if op is None else {"op": op}
        if not self.host:
            self.dist.all_reduce(t, **kw)
            return
        h = t.cpu()
        self.dist.all_reduce(h, **kw)
        t.copy_(h)


def roofline_blocks(phase_tot, args, dims, world, dt, value):
    """the `roofline` object of the JSON line for the default (int8 MAC) build, from rank 0's phase totals (HIP events on the library's queue)"""
    n_ind, m_snp, nbr_x, mct_x, N, L, D, SLOTS, LEVEL = dims
    ctw, outw = 2 * (LEVEL + 1) * N, 2 * L * N
    ms_small, n_small, by_small = phase_tot.get("mac_small", [0.0, 0, 0.0])
    ms_ntt, n_ntt, by_ntt = phase_tot.get("ntt_plain", [0.0, 0, 0.0])
    n_ntt_all = phase_tot.get("ntt_plain_all", [0.0, 0, 0.0])[1]
    ntt_avg_ms = ms_ntt / max(n_ntt, 1)
    ntt_total_ms = ntt_avg_ms * n_ntt_all
    bytes_per_plain = (N // 2) * (8 + 5 * (L - 1) + 6)                   # coefficient row in; five digit planes per 35-bit modulus and six for the 46-bit one out (encode.hip credits the same)
    plains_per_launch = (by_ntt / max(n_ntt, 1)) / bytes_per_plain
    NTT_FP64_INSTR = 2016                                               # fp64 vector instructions per thread of k_ntt_half3 (static count of the gfx950 ISA)
    ntt_instr_s = plains_per_launch * L * 256 * NTT_FP64_INSTR / (ntt_avg_ms * 1e-3) if n_ntt else 0.0
    ntt_blk = {"bound": "valu_fp64", "achieved": 2.0 * ntt_instr_s / 1e12, "peak": 2.0 * FP64_VALU_SPEC_FMA_S / 1e12, "unit": "TFLOP/s",
               "frac": ntt_instr_s / FP64_VALU_SPEC_FMA_S, "frac_kind": "fp64 ISSUE-SLOT fraction: every fp64 vector instruction (mul, rndne, fma, add) counts as one slot "
               "of 64 lanes; 'TFLOP/s' here is slots x 64 lanes x 2 (FMA-equivalent) so that it compares with the guide's fp64 vector peak - it is not a count of "
               "floating-point operations performed", "kernel": "k_ntt_half3<false, true>", "avg_launch_ms": ntt_avg_ms, "launches": n_ntt_all,
               "launches_timed": n_ntt, "total_ms_in_timed_region": ntt_total_ms,
               "what": "plaintext (panel) NTT: 2016 fp64 vector instructions per thread and (plaintext, modulus) row (13 stages x 16 butterflies x 8 + the degenerate first "
                       "stage + canonicalisation), one issue slot = 64 lanes counted as 2 flop (FMA-equivalent; the mix is mul, rndne, fma, add) against the fp64 vector peak "
                       "256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz",
               "hbm": {"alg_bytes_per_launch": by_ntt / max(n_ntt, 1), "achieved_GBps": (by_ntt / max(ms_ntt, 1e-9)) / 1e6, "peak_GBps": HBM_PEAK_GBS,
                       "frac": (by_ntt / max(ms_ntt, 1e-9)) / 1e6 / HBM_PEAK_GBS}}
    mac_gbps = by_small / (ms_small * 1e-3) / 1e9
    padded_macs_s = 2 * nbr_x * mct_x * D * D * 2 * KP * (L - 1) * N * args.steps / world / (ms_small * 1e-3)
    refetch, refetch_src = None, None
    try:        # exact fabric-side read bytes of this kernel (TCC_EA0_RDREQ_{32B,64B,128B}) against its operand tiles: profiles/r04_pmc_mac_i8_ring.json
        pm4 = json.load(open(os.path.join(ROOT, "profiles", "r04_pmc_mac_i8_ring.json")))
        refetch = pm4["default"]["k_mac_i8"]["read_bytes_per_launch"] / pm4["_algorithmic_read_bytes_K1183"]["total"]
        refetch_src = "profiles/r04_pmc_mac_i8_ring.json (static: counter passes of k_mac_i8_ring<5, 3, 0, 2> at c2, K = 1183)"
    except Exception:
        pass
    mac_blk = {"bound": "hbm", "achieved": mac_gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": mac_gbps / HBM_PEAK_GBS, "kernel": "k_mac_i8_ring<5, 3>",
               "frac_of_achievable_6300": mac_gbps / 6300.0, "read_bytes_over_operand_bytes": refetch, "read_bytes_source": refetch_src,
               "avg_launch_ms": ms_small / n_small, "launches": n_small, "total_ms_in_timed_region": ms_small, "alg_bytes_per_launch": by_small / n_small,
               "padded_ring_macs_per_s_in_kernel": padded_macs_s, "int8_macs_per_s_in_kernel": 25.0 * padded_macs_s * (32 * 96) / (30 * 91),
               "what": "ring MAC of the four 35-bit moduli on v_mfma_i32_16x16x64_i8: operands as five signed base-256 digits, 25 digit products per ring-MAC into nine "
                       "int32 sums, Horner mod q in the epilogue (exact); both k-contiguous digit streams prefetched global -> LDS by the DMA engine two chunks ahead "
                       "(three 50 KiB slots), read once; bytes = the two streams + tile-ordered results written.  The 46-bit modulus runs the same kernel with six digits "
                       "(k_mac_i8_ring<6, 2>, phase mac_big)",
               "helpers_ms_per_step": {k: phase_tot[k][0] / args.steps for k in ("mac_i8_pack_pt", "mac_i8_pack_rot", "mac_i8_untile") if k in phase_tot}}
    # The riding transposition: most plaintext-NTT launches carry mover workgroups that transpose the previous MAC launch's plaintext panel into
    # the int8 MAC's tiles - ONE kernel, k_ntt_half3_move, doing an fp64-issue-bound job and an HBM-bound job side by side.  Its algorithmic bytes are the NTT's plus
    # the movers' (1 B in + 1 B out per digit byte); it is priced against HBM (the closer roof), with the NTT's issue-slot fraction inside that launch beside it.
    ms_rd, n_rd, by_rd_ntt = phase_tot.get("ntt_plain_ride", [0.0, 0, 0.0])
    n_rd_all = phase_tot.get("ntt_ride_all", [0.0, 0, 0.0])[1]
    by_mv = phase_tot.get("pt_ride", [0.0, 0, 0.0])[2]
    ride_blk, ride_total_ms = None, 0.0
    if n_rd and n_rd_all:
        rd_avg_ms = ms_rd / n_rd
        ride_total_ms = rd_avg_ms * n_rd_all
        rd_bytes = (by_rd_ntt + by_mv) / n_rd
        rd_gbps = rd_bytes / (rd_avg_ms * 1e-3) / 1e9
        rd_plains = (by_rd_ntt / n_rd) / bytes_per_plain
        rd_instr_s = rd_plains * L * 256 * NTT_FP64_INSTR / (rd_avg_ms * 1e-3)
        ride_blk = {"bound": "hbm", "achieved": rd_gbps, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rd_gbps / HBM_PEAK_GBS, "kernel": "k_ntt_half3_move<false, 1, true>",
                    "avg_launch_ms": rd_avg_ms, "launches": n_rd_all, "launches_timed": n_rd, "total_ms_in_timed_region": ride_total_ms,
                    "alg_bytes_per_launch": rd_bytes, "alg_bytes_ntt": by_rd_ntt / n_rd, "alg_bytes_movers": by_mv / n_rd,
                    "fp64_issue": {"frac": rd_instr_s / FP64_VALU_SPEC_FMA_S, "what": "the NTT workgroups' fp64 issue slots over the same launch duration (2016 per thread and row)"},
                    "plain_launch": {"kernel": "k_ntt_half3<false, true>", "avg_launch_ms": ntt_avg_ms, "launches": n_ntt_all, "fp64_issue_frac": ntt_blk["frac"],
                                     "what": "the same NTT without mover workgroups (the encodes no delayed MAC launch rides in: the first of a call and of every block-row group)"},
                    "what": "plaintext (panel) NTT with the riding transposition: per launch 2048 plaintexts x 5 moduli of half-size NTTs (fp64 issue bound: 64 KiB in, "
                            "26 B x N/2 out per plaintext) and, on 192 mover workgroups dispatched first (one per CU, beside three NTT workgroups), 1/launches of the previous "
                            "MAC launch's [k][coefficient] -> [coefficient][16 k] byte transposition (HBM bound: 1 B in + 1 B out per digit byte).  The pair shares the memory "
                            "system: priced against the HBM peak with both jobs' algorithmic bytes"}
    if ride_blk and ride_total_ms >= ms_small:
        dom, other = ride_blk, mac_blk
    else:
        dom, other = (ntt_blk, mac_blk) if ntt_total_ms >= ms_small else (mac_blk, ntt_blk)
    alg_step = 2 * n_ind * m_snp + (KP * nbr_x + KP * mct_x) * ctw * 8 + (KP * mct_x + KP * nbr_x) * outw * 8
    hbm_alg = alg_step * args.steps / dt / 1e9
    traffic, traffic_src = None, "no counter pass of this kernel in profiles/"
    try:
        for name in ("r06_pmc_ntt_ride.json", "r05_pmc_ntt.json", "r03_pmc_traffic_per_launch_i8.json"):
            path = os.path.join(ROOT, "profiles", name)
            if not os.path.exists(path):
                continue
            pm = json.load(open(path))
            ks = [k for k in pm if dom["kernel"].split("<")[0] in k and not k.startswith("_")]
            if ks:
                best = max(ks, key=lambda k: pm[k]["launches"])
                traffic = pm[best]["hbm_bytes_per_launch"]
                traffic_src = (f"profiles/{name} (static: separate rocprofv3 --pmc passes at config {pm.get('_config')}; "
                               "the NTT's launch shape - 2048 plaintexts x 5 moduli - is the same at every config; the riding movers' share per launch is the "
                               "MAC group's, i.e. that of the config the passes ran at)")
                break
    except Exception:
        pass
    return dict(dom, traffic=traffic, traffic_source=traffic_src, second_kernel=other,
                hbm_algorithmic={"bytes_per_step": alg_step, "achieved_GBps": hbm_alg, "frac": hbm_alg / HBM_PEAK_GBS,
                                 "what": "SURVEY §8(d): int8 genotypes once per product + ciphertexts in/out, divided by the WHOLE step time"})


PHASES = ("rotate", "skew", "encode", "mac", "mac_small", "mac_big", "ntt_plain", "ntt_plain_all", "ntt_plain_ride", "ntt_ride_all", "pt_ride", "mac_i8_pack_pt", "mac_i8_pack_rot",
          "mac_i8_untile")


class Watchdog:
    """A bounded wait around calls that only return when every rank has entered them (ncclCommInitRank, the first collectives): after `seconds` the process prints
    what it was waiting in and ends with exit code 3 (os._exit: the stuck thread cannot be interrupted).  Never a re-exec - the GPU box forbids that."""

    def __init__(self, seconds, what):
        import threading
        self.t = threading.Timer(seconds, self._fire, (seconds, what))
        self.t.daemon = True
        self.t.start()

    @staticmethod
    def _fire(seconds, what):
        print(f"[bench] WATCHDOG: {what} did not finish within {seconds:.0f} s - a rank is missing or the fabric is stuck; exiting with code 3", file=sys.stderr, flush=True)
        os._exit(3)

    def cancel(self):
        self.t.cancel()


# SHA-256 digests of both products per config (inputs are seeded, so they are constants of the repository; equal for every world size).  c4 is what
# tests/test_gpu_fullsize.py pins together with oracle comparisons at c4's launch shapes; c2 / c3 are the single-GPU lines of tests/test_gpu_multirank.py.
PINNED_DIGESTS = {
    "c4": ("cab05b5a8326ff9dc51f0e139c2261d47dc2a8830541a134273bffbc6f3888e6", "ce9b0e28cb6318dafb77b27fbdff1d4491f3fcb3e70548d5bfac723753d7ee62"),
    "c3": ("75244cd9836057918b53e584e72004d7aee63a5908e80a2e8a1181730467718a", "4e072d7c88b672eb61068ff02e799c17f92b5a58da579c42e12d9329ff3591d8"),
    "c2": ("fbececb714c4aab7ab983ebc25838a00ec5c24ff123268c9c4999b645f3d37d3", "06753f44a4a7795161f6abe191e28332b0e3db0d127c73c5e93d0591cbbd6672"),
}


def digests_match_pinned(config, digests):
    """True / False; None when the config has no pinned value"""
    pin = PINNED_DIGESTS.get(config)
    if not pin or not digests:
        return None
    return (digests["out1_sha256"], digests["out2_sha256"]) == pin


def main_lib_engine(args):
    """N > 1 with the multi-GPU sequence INSIDE the library (sfg_mgpu_*, sfgwas_amd/csrc/mgpu.hip): this script only makes the synthetic inputs, calls the two
    products per step, keeps the clock and hashes the outputs.  One process per GPU (the driver's launch: each rank joins with sfg_mgpu_create_rank and a 128-byte
    id passed through torch.distributed's TCP store) or --single-process (sfg_mgpu_create: one host thread per device, the Go party's form)."""
    solo = os.environ.get("SFG_MGPU_SOLO")            # r/w: TIMING ONLY - this process runs rank r of a w-rank world on one GPU, the exchanges stood in by local copies
    single = args.single_process or (args.gpus == 1 and "RANK" not in os.environ) or bool(solo)
    if solo:
        args.no_check = args.no_digest = True
    if not single and "RANK" not in os.environ:           # no launcher around us: start the N ranks as children (never exec after touching the GPU)
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            port = so.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MASTER_ADDR="127.0.0.1")
        sys.exit(subprocess.call(cmd, env=env))
    import numpy as np
    import torch
    import torch.distributed as dist
    from sfgwas_amd import capi
    from sfgwas_amd import params as P

    SLOTS, D, N, L, LEVEL = P.SLOTS, P.D, P.N, P.MAX_LEVEL, P.MAX_LEVEL
    world = args.gpus
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if single:
        rank, local_rank = 0, 0
        devices = [int(x) for x in args.devices.split(",")] if args.devices else list(range(world))
        if len(devices) != world:
            raise SystemExit(f"--devices names {len(devices)} ranks, --gpus {world}")
        watchdog = Watchdog(float(os.environ.get("SFG_BENCH_COMM_TIMEOUT_S", "300")), "communicator creation (ncclCommInitAll) / pre-flight collectives")
        mg = capi.MultiGpu(P.Q_PN14, P.P_PN14, devices=devices)
    else:
        rank, local_rank = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
        if int(os.environ["WORLD_SIZE"]) != world:
            raise SystemExit(f"--gpus {world} but WORLD_SIZE={os.environ['WORLD_SIZE']}")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)          # host-side only: the id, barriers, the clock's max, the digest gather.  The data path's
        devices = [local_rank]                                                # collectives are the library's own RCCL calls.

        def agree(ok, what):
            """every rank leaves with the same verdict (MIN over gloo) BEFORE anybody enters a call that only returns when all ranks have entered it"""
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                if args.allow_engine_fallback:
                    if rank == 0:
                        print(f"[bench] {what} failed on a rank: falling back to --engine torch (--allow-engine-fallback)", file=sys.stderr, flush=True)
                    dist.destroy_process_group()
                    return False
                raise SystemExit(f"[bench] rank {rank}: {what} failed on {'this' if not ok else 'another'} rank; no fallback without --allow-engine-fallback")
            return True
        # (1) can every rank make a context on its device at all?  ncclCommInitRank below returns only when ALL ranks have called it: a rank that dies before it
        # would leave the others waiting for ever, so the cheap failure modes are found and agreed on first.
        err = None
        try:
            probe = capi.Context(P.Q_PN14, P.P_PN14, device=local_rank)
            probe.close()
        except Exception as e:                                               # noqa: BLE001 - whatever it is, the other ranks must hear of it
            err = e
            print(f"[bench] rank {rank}: cannot create a context on device {local_rank}: {e}", file=sys.stderr, flush=True)
        if not agree(err is None, "sfg_ctx_create"):
            return "fallback"
        box = [capi.MultiGpu.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        # (2) communicator creation and the first collectives under a watchdog: RCCL has no timeout of its own.  The watchdog ends THIS process (never a re-exec)
        watchdog = Watchdog(float(os.environ.get("SFG_BENCH_COMM_TIMEOUT_S", "300")), f"rank {rank}: communicator creation / pre-flight collectives")
        try:
            mg = capi.MultiGpu(P.Q_PN14, P.P_PN14, rank=rank, world=world, uid=box[0], device=local_rank)
            ok = True
        except capi.SfgError as e:
            print(f"[bench] rank {rank}: sfg_mgpu_create_rank failed: {e}", file=sys.stderr, flush=True)
            mg, ok = None, False
        if not agree(ok, "sfg_mgpu_create_rank"):
            if mg is not None:
                mg.close()
            watchdog.cancel()
            return "fallback"
    # (3) pre-flight: a reduce-scatter and an all-reduce of a known pattern through the engine's own exchange functions, checked on the host (sfg_mgpu_preflight);
    # what the communicator itself says about the world (ncclCommCount) goes into the line
    mg.preflight(1 << 16)
    comm_ranks = [mg.comm_info(i)[0] for i in range(mg.nlocal)]
    watchdog.cancel()
    if not single:
        allr = [None] * world
        dist.all_gather_object(allr, comm_ranks)
        comm_ranks = [x for sub in allr for x in sub]
    lib = capi.lib()
    nloc = mg.nlocal
    shared_device = len(set(devices)) < len(devices)
    mg.fill_rotkeys_synthetic(P.rotations_for_matmul(), 0xBEEF)
    n_ind, m_snp = CONFIGS[args.config]
    nbr_x, mct_x = ceil_div(n_ind, SLOTS), ceil_div(m_snp, SLOTS)
    gate = None
    if rank == 0 and not args.no_check:
        gate = parity_gate(mg.ctx[0], capi, P)
    g = mg.geno_synthetic(n_ind, m_snp, 0x5F6A, packed=args.packed_geno)
    blocks = [mg.geno_blocks(g, i) for i in range(nloc)]
    A1, A2, out1, out2 = [], [], [], []
    for i, c in enumerate(mg.ctx):
        b0, b1 = blocks[i]
        a1 = capi.DevArray(c, (KP, nbr_x, 2, LEVEL + 1, N))                 # Q : kp x n_ind (replicated)
        c.check(lib.sfg_fill_uniform_ct_dev(c.h, a1.p, KP * nbr_x, LEVEL, 0xC1F3), "fill A1")
        a2 = capi.DevArray(c, (KP, max(b1 - b0, 1), 2, LEVEL + 1, N))       # Q': kp x m_snp, this rank's SNP blocks; ciphertext (i, global block b) has seed base + i*mct_x + b
        for r in range(KP):
            if b1 > b0:
                c.check(lib.sfg_fill_uniform_ct_dev(c.h, C.c_void_p(a2.p.value + r * (b1 - b0) * 2 * (LEVEL + 1) * N * 8), b1 - b0, LEVEL, 0xD2A7_0000 + r * mct_x + b0), "fill A2")
        A1.append(a1); A2.append(a2)
        out1.append(capi.DevArray(c, (KP, max(b1 - b0, 1), 2, L, N)))
        out2.append(capi.DevArray(c, (KP, nbr_x, 2, L, N)))
    mg.sync()
    ctx0 = mg.ctx[0]
    # Every rank's library queue is an explicit torch stream, as in the N = 1 path and the torch engine: the configuration every single-GPU number of this repository
    # was measured in.  (Measured in round 5, profiles/r05_mgpu_queue_count.txt: with the product on the context's OWN queue and a fourth library stream in use
    # every kernel started 15 - 30 us later; SFG_BENCH_OWN_STREAM=1 keeps the library's own queues.)
    _streams = []
    if os.environ.get("SFG_BENCH_OWN_STREAM") != "1":
        for i, c in enumerate(mg.ctx):
            st = torch.cuda.Stream(device=torch.device("cuda", devices[i]))
            c.check(lib.sfg_ctx_set_stream(c.h, C.c_void_p(st.cuda_stream)), "set_stream")
            _streams.append(st)
    phase_tot = {}

    def add_phases():
        for ph in PHASES:
            ms = ctx0.phase_ms(ph)
            if ms >= 0:
                n = lib.sfg_last_phase_launches(ctx0.h, ph.encode())
                a = phase_tot.setdefault(ph, [0.0, 0, 0.0])
                a[0] += ms
                a[1] += n
                a[2] += max(lib.sfg_last_phase_bytes(ctx0.h, ph.encode()), 0.0)

    def step():
        lib.sfg_ctx_clear_phases(ctx0.h)
        mg.matmul_dev(A1, KP, LEVEL, L, g, 0, out1)                         # Q * X   : the ranks' own output block columns
        mg.matmul_dev(A2, KP, LEVEL, L, g, capi.SFG_TRANSPOSE, out2)        # Q' * X^T: contraction over the ranks' SNP blocks + the exchange
        add_phases()                                                        # (resolves rank 0's timing events: one host wait per step, after both products are enqueued)

    def barrier():
        mg.sync()
        if not single:
            dist.barrier()
        for dv in set(devices):
            torch.cuda.synchronize(dv)

    pt_cache = {}

    def enable_pt_cache():
        want = os.environ.get("SFG_BENCH_PT_CACHE_GB", "auto")
        if want == "0" or (want == "auto" and shared_device):
            return
        barrier()
        budget = min(torch.cuda.mem_get_info(dv)[0] for dv in set(devices)) - (16 << 30)
        if want != "auto":
            budget = min(budget, int(float(want) * (1 << 30)))
        if budget >= (1 << 29):
            mg.check(lib.sfg_mgpu_geno_set_plaintext_cache(mg.h, g, C.c_size_t(budget)), "plaintext cache")
            pt_cache["budget"] = budget

    for w in range(args.warmup):
        step()
        if w == 0:
            enable_pt_cache()
    phase_tot.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if not single:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    near_ties = ctx0.encoder_near_ties()
    unprovable = C.c_ulonglong()
    ctx0.check(lib.sfg_ctx_encoder_unprovable(ctx0.h, C.byref(unprovable)), "encoder_unprovable")
    if "budget" in pt_cache:
        v = [C.c_size_t() for _ in range(4)]
        ctx0.check(lib.sfg_geno_plaintext_cache_stats(ctx0.h, C.c_void_p(lib.sfg_mgpu_geno_shard(g, 0)), *[C.byref(x) for x in v]), "plaintext cache stats")
        pt_cache.update(blocks=v[0].value, bytes=v[1].value, hits=v[2].value, fills=v[3].value)
    digests = None
    if not args.no_digest:
        def ct_hashes(h):                                   # [KP][ncols][...] -> bytes [KP][ncols][32]
            return np.frombuffer(b"".join(hashlib.sha256(h[i, j].tobytes()).digest() for i in range(h.shape[0]) for j in range(h.shape[1])),
                                 dtype=np.uint8).reshape(h.shape[0], h.shape[1], 32)
        parts = [(blocks[i][0], ct_hashes(out1[i].host())) for i in range(nloc) if blocks[i][1] > blocks[i][0]]
        if not single:
            allp = [None] * world
            dist.all_gather_object(allp, parts)
            parts = [p for sub in allp for p in sub]
        parts.sort(key=lambda p: p[0])
        if rank == 0:
            o2 = ct_hashes(out2[0].host())
            for i in range(1, nloc):                        # every rank holds the complete Q'X^T: they must agree
                if not np.array_equal(ct_hashes(out2[i].host()), o2):
                    raise SystemExit(f"rank {mg.ranks[i]} holds a different Q'*X^T than rank 0")
            digests = {"out1_sha256": hashlib.sha256(np.concatenate([p[1] for p in parts], axis=1).tobytes()).hexdigest(),
                       "out2_sha256": hashlib.sha256(o2.tobytes()).hexdigest(),
                       "of": "SHA-256 over the per-ciphertext SHA-256s in [i][j] order: Q*X (kp x m_ct) and Q'*X^T (kp x nbr); equal for every world size"}
    macs_per_step = 2 * n_ind * m_snp * KP * 2 * L * (N // SLOTS)
    value = macs_per_step * args.steps / dt
    if solo:
        print(json.dumps({"solo_rank_timing_only": f"rank {mg.ranks[0]} of {mg.world} (library engine, SFG_MGPU_SOLO)", "config": args.config, "ms_per_step": 1e3 * dt / args.steps,
                          "kernel_phases_ms_per_step": {k: v[0] / args.steps for k, v in phase_tot.items()}, "plaintext_cache": pt_cache,
                          "note": "exchanges replaced by local copies: the outputs are not a product"}), flush=True)
        for a in A1 + A2 + out1 + out2:
            a.free()
        mg.geno_free(g)
        mg.close()
        return
    if rank == 0:
        res = {
            "metric": "pca_power_iter_ring_macs_per_s", "value": value, "unit": "ring-MAC/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64 ring words, exact integers: int8 digits on the matrix core for the MAC of all five moduli, fp64-held integers in the encode and key-switch kernels",
            "data": "synthetic",
            "config": {"workload": f"{args.config}: one PCA power iteration local work = Q*X + Q'*X^T, {n_ind} x {m_snp} int8 genotypes, "
                                   f"kp={KP}, PN14QP438-shaped ring (N=16384, L=5 of 6 moduli), on-the-fly diagonal encode",
                       "parallelism": f"snp-block x{world}", "power_iter_wall_s": dt / args.steps,
                       "genotype_residency": "2-bit packed (sfg_geno_pack)" if args.packed_geno else "int8",
                       "engine": "libsfgwas_hip sfg_mgpu_* (mgpu.hip): " + ("one process, one host thread per device" if single else "one process per GPU, joined by sfg_mgpu_create_rank"),
                       "collectives": {"rccl": "RCCL (called by the library; librccl resolved with dlopen)",
                                       "direct": "in-process direct transport (ranks share a device: a rehearsal; timing not meaningful)"}.get(mg.transport, mg.transport),
                       "plaintext_cache": (f"{pt_cache['blocks']} blocks of rank 0 ({pt_cache['bytes'] / 2**30:.1f} GiB; {pt_cache.get('hits', 0)} block encodes served from it, "
                                           f"{pt_cache.get('fills', 0)} filled)" if "budget" in pt_cache else "off"),
                       "rotation_cache_QX": "replicated",
                       "QtXt_reduce_scatter": "per output block column, on the collectives' queue beside the next column's product"},
        }
        # what a multi-GPU record must be able to say by itself: how many ranks RCCL saw, which engine ran, and whether the words are the pinned ones
        res["rccl_ranks"] = comm_ranks[0] if len(set(comm_ranks)) == 1 else comm_ranks
        res["rccl_ranks_of"] = "ncclCommCount of every rank's communicator (sfg_mgpu_comm_info); 0 = no communicator (direct transport / world 1)"
        res["engine_fallback"] = False
        res["preflight"] = "ok: reduce-scatter + all-reduce of a known uint64 pattern through the engine's exchange functions, 65536 words per rank slice, checked on the host"
        if gate is not None:
            res["parity_gate"] = gate
        if digests is not None:
            res["digests"] = digests
            res["digests_match_pinned"] = digests_match_pinned(args.config, digests)
        res["encoder_near_ties"] = {"count": near_ties, "within_2^-50": unprovable.value, "what": "rank 0; see the N = 1 line"}
        if phase_tot.get("mac_small", [0, 0, 0])[1] and "mac_i8_pack_pt" in phase_tot:
            res["roofline"] = roofline_blocks(phase_tot, args, (n_ind, m_snp, nbr_x, mct_x, N, L, D, SLOTS, LEVEL), world, dt, value)
            res["roofline"]["of"] = "rank 0's launches (HIP events on its queue)"
        res["phases_ms_per_step"] = {k: v[0] / args.steps for k, v in phase_tot.items()}
        print(json.dumps(res), flush=True)
    for a in A1 + A2 + out1 + out2:
        a.free()
    mg.geno_free(g)
    mg.close()
    if not single:
        dist.destroy_process_group()
    if gate is not None and gate["status"] != "ok":
        raise SystemExit("parity gate FAILED: the HIP path differs from the oracle")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", default=os.environ.get("SFG_BENCH_CONFIG", "c4"))
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="bounded CPU-baseline sample (seconds of wall time)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-end-to-end", action="store_true", help="skip the end-to-end CPU leg (one 8192 x 8192 block through the whole CPU restatement, ~ 30 s)")
    ap.add_argument("--no-check", action="store_true", help="skip the oracle parity gate")
    ap.add_argument("--no-digest", action="store_true", help="skip the SHA-256 digests of the outputs")
    ap.add_argument("--backend", default=os.environ.get("SFG_BENCH_BACKEND", "nccl"), choices=("nccl", "gloo"),
                    help="nccl = RCCL, one GPU per rank (the measured path); gloo = host-staged collectives, ranks may share one GPU (rehearsal)")
    ap.add_argument("--packed-geno", action="store_true", help="keep the genotype matrix 2-bit packed in HBM (sfg_geno_pack: 4x smaller, blocks expanded on the fly)")
    ap.add_argument("--engine", default=os.environ.get("SFG_BENCH_ENGINE", "lib"), choices=("lib", "torch"),
                    help="N > 1 only.  lib (default): the multi-GPU sequence runs INSIDE libsfgwas_hip (sfg_mgpu_*: SNP-block shards, per-column reduce-scatter over RCCL "
                         "on a second queue, finalize of the owned giants, all-reduce); torch: the same sequence issued from this script through torch.distributed (the A/B baseline)")
    ap.add_argument("--single-process", action="store_true",
                    help="engine lib: ONE process drives all N GPUs (sfg_mgpu_create + ncclCommInitAll, one host thread per device) - the form a Go party process uses - "
                         "instead of one process per GPU")
    ap.add_argument("--allow-engine-fallback", action="store_true",
                    help="N > 1, engine lib: if the library's engine cannot be created on some rank, run the torch-issued sequence instead (the line then says "
                         "engine_fallback: true).  Default: exit non-zero with the failing rank's error - a scaling record must not change what it measures silently")
    ap.add_argument("--devices", default=None, help="--single-process: comma-separated device indices of the N ranks (a repeated device selects the in-process "
                                                    "'direct' transport: a rehearsal of N > 1 on one GPU; timings then mean nothing)")
    args = ap.parse_args()
    # switches of this script that need more than the product library offers: the timing-only solo rank and the kernel A/B switches live in the experimenters' build
    # (sfgwas_amd/lib_ab, `make -C sfgwas_amd/csrc ab`); the forced exchange at one rank is a test switch of the product library
    if os.environ.get("SFG_MGPU_SOLO") and not os.environ.get("SFG_LIB_PATH"):
        from sfgwas_amd import capi as _capi
        os.environ["SFG_LIB_PATH"] = _capi.ab_lib()
        _capi.LIB_PATH = os.environ["SFG_LIB_PATH"]
    if os.environ.get("SFG_MGPU_FORCE_COLLECTIVES") == "1":
        os.environ.setdefault("SFG_ENABLE_TEST_HOOKS", "1")
    # SFG_MGPU_FORCE_COLLECTIVES=1 runs the library's exchange even with one rank (over RCCL): `--gpus 1` then takes the engine too
    if ((args.gpus > 1 or os.environ.get("SFG_MGPU_FORCE_COLLECTIVES") == "1" or os.environ.get("SFG_MGPU_SOLO")) and args.engine == "lib" and args.backend == "nccl"
            and not os.environ.get("SFG_BENCH_SOLO") and os.environ.get("SFG_BENCH_FORCE_COLLECTIVES") != "1"):
        rc = main_lib_engine(args)
        if rc != "fallback":
            return rc
        args.engine = "torch"             # --allow-engine-fallback: the library's engine could not be created on this node (every rank agreed): the torch-issued sequence
        args.engine_fallback = True
    return main_torch(args)


def main_torch(args):

    # `python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as CHILD processes (one per GPU, torch.distributed.run) before
    # anything in this process touches torch.cuda / HIP, relay rank 0's JSON line and exit with the launcher's code.  (No exec: a process that has
    # initialised the GPU must not be replaced, and this parent never initialises it.)
    if args.gpus > 1 and "RANK" not in os.environ and not os.environ.get("SFG_BENCH_SOLO"):
        import socket
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), MASTER_ADDR="127.0.0.1")
        sys.exit(subprocess.call(cmd, env=env))

    import numpy as np
    import torch
    import torch.distributed as dist
    from sfgwas_amd import capi
    from sfgwas_amd import params as P
    from sfgwas_amd.sharding import snp_block_range, giant_slots

    SLOTS, D, N, L, LEVEL = P.SLOTS, P.D, P.N, P.MAX_LEVEL, P.MAX_LEVEL
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # SFG_BENCH_SOLO=r/w: ONE process computes the share of rank r of a w-rank run, collectives replaced by local copies of its own slices.
    # TIMING ONLY (the outputs are not a product): per-rank phase times of world sizes this pool cannot run, for the model in DESIGN.md §6.
    solo = None
    if os.environ.get("SFG_BENCH_SOLO"):
        solo = tuple(int(x) for x in os.environ["SFG_BENCH_SOLO"].split("/"))
        rank, world = solo
        args.no_check = args.no_digest = args.no_cpu_baseline = True
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    if args.backend == "gloo":
        local_rank %= torch.cuda.device_count()            # rehearsal: ranks share the GPUs that exist
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # SFG_BENCH_FORCE_COLLECTIVES=1 runs the collectives even at world size 1 (under torch.distributed.run)
    force_coll = os.environ.get("SFG_BENCH_FORCE_COLLECTIVES") == "1"
    use_dist = world > 1 or (force_coll and "RANK" in os.environ)
    coll = None
    if solo:
        coll = Coll(dist, torch, "solo", solo)
    elif use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(args.backend, rank=rank, world_size=world)
        coll = Coll(dist, torch, args.backend)
    # Q*X's rotation cache: every rank rebuilds it (default) or the ranks build 1/world each and all-gather (SFG_BENCH_ROTCACHE=sharded).  Measured at
    # 100k x 1M (tools/r3_solo.sh, profiles/r03_solo_rank_phases_c4.jsonl): the whole build is 72 ms of key switching per rank, a 1/8 shard + scatter 19 ms,
    # and the all-gather that replaces the difference moves 25 GB into every rank - more than 53 ms on 7 xGMI links.  Identical digests either way.
    shard_rotcache = use_dist and os.environ.get("SFG_BENCH_ROTCACHE", "replicated") == "sharded"

    n_ind, m_snp = CONFIGS[args.config]
    nbr_x, mct_x = ceil_div(n_ind, SLOTS), ceil_div(m_snp, SLOTS)          # block rows / cols of X
    blk0, blk1, c0, c1 = snp_block_range(m_snp, rank, world)                # SNP-block shard of this rank
    m_loc, nblk_loc = c1 - c0, blk1 - blk0

    ctx = capi.Context(P.Q_PN14, P.P_PN14, device=local_rank)
    lib = capi.lib()
    # One explicit (non-default) stream carries torch's tensor ops, the library's launches and - through torch.distributed's stream
    # synchronisation - the RCCL collectives.  torch's DEFAULT stream has the handle 0, which sfg_ctx_set_stream reads as "use the context's own
    # stream": the collectives would then not be ordered after the library's kernels.
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    assert stream.cuda_stream != 0

    def chk(rc, what):
        ctx.check(rc, what)

    if (use_dist and not solo) or os.environ.get("SFG_BENCH_OWN_STREAM") != "1":     # (SFG_BENCH_OWN_STREAM=1, one rank: the library keeps its own queues - the CU-partitioning experiments mask THOSE)
        chk(lib.sfg_ctx_set_stream(ctx.h, C.c_void_p(stream.cuda_stream)), "set_stream")     # a refused handle would leave the collectives unordered against the MAC

    rots = P.rotations_for_matmul()
    arr = (C.c_int * len(rots))(*rots)
    chk(lib.sfg_fill_rotkeys_synthetic(ctx.h, arr, len(rots), 0xBEEF), "fill rotkeys")

    gate = None
    if rank == 0 and not args.no_check:
        gate = parity_gate(ctx, capi, P)
    # ---- synthetic inputs, generated on the device (seeds: BASELINE.md §2); windows of ONE global matrix / ciphertext grid
    geno = torch.empty((n_ind, m_loc), dtype=torch.int8, device=dev)
    chk(lib.sfg_fill_geno_window_dev(ctx.h, C.c_void_p(geno.data_ptr()), n_ind, m_loc, m_loc, c0, m_snp, 0x5F6A), "fill_geno")
    gh = C.c_void_p()
    chk(lib.sfg_geno_from_device(ctx.h, C.c_void_p(geno.data_ptr()), n_ind, m_loc, m_loc, C.byref(gh)), "geno_from_device")
    if args.packed_geno:
        gp = C.c_void_p()
        chk(lib.sfg_geno_pack(ctx.h, gh, C.byref(gp)), "geno_pack")
        lib.sfg_geno_free(ctx.h, gh)
        gh = gp
        del geno
        torch.cuda.empty_cache()
    ctw = 2 * (LEVEL + 1) * N
    A1 = torch.empty((KP, nbr_x, ctw), dtype=torch.int64, device=dev)       # Q   : kp x n_ind (replicated)
    A2 = torch.empty((KP, nblk_loc, ctw), dtype=torch.int64, device=dev)    # Q'  : kp x m_snp, this rank's SNP blocks
    chk(lib.sfg_fill_uniform_ct_dev(ctx.h, C.c_void_p(A1.data_ptr()), KP * nbr_x, LEVEL, 0xC1F3), "fill A1")
    for i in range(KP):                                                     # ciphertext (i, global block b) has seed base + i*mct_x + b
        chk(lib.sfg_fill_uniform_ct_dev(ctx.h, C.c_void_p(A2[i].data_ptr()), nblk_loc, LEVEL, 0xD2A7_0000 + i * mct_x + blk0), "fill A2")
    outw = 2 * L * N
    out1 = torch.empty((KP, nblk_loc, outw), dtype=torch.int64, device=dev)
    out2 = torch.empty((KP, nbr_x, outw), dtype=torch.int64, device=dev)
    gpr, g_lo, g_hi = giant_slots(rank, world)                              # giants per rank (padded), this rank's giant range
    col = D * KP * outw                                                     # accumulator words of one output block column [giant][i][2][L][N]
    colp = world * gpr * KP * outw                                          # the same padded to world * gpr giant slots (reduce-scatter window)
    acc_mine = torch.empty((nbr_x, gpr, KP, outw), dtype=torch.int64, device=dev) if use_dist else None
    # ---- rotation caches as objects (N > 1): Q*X's cache is built in shards and gathered; Q'*X^T's (this rank's own block rows) is built in place so
    # that the product can run one output block column at a time beside the previous column's reduce-scatter
    jobw, tailw = C.c_size_t(), C.c_size_t()
    if use_dist:
        chk(lib.sfg_rotcache_layout(ctx.h, KP, L, C.byref(jobw), C.byref(tailw)), "rotcache_layout")
    jobw, tailw = jobw.value, tailw.value
    njobs = nbr_x * KP
    jpr = ceil_div(njobs, world)                                            # jobs per rank (the last rank's range may be short: padded)
    job0, job1 = min(rank * jpr, njobs), min((rank + 1) * jpr, njobs)
    cache1_w = njobs * jobw + tailw if shard_rotcache else 0
    cache2_w = nblk_loc * KP * jobw + tailw
    cache2_budget = float(os.environ.get("SFG_BENCH_CACHE2_GB", "72")) * (1 << 30)
    pipe_cols = use_dist and cache2_w * 8 <= cache2_budget                  # else: the library's own grouped rotation cache, collectives afterwards
    if not pipe_cols:
        cache2_w = 0
    cache_buf = torch.empty(max(cache1_w, cache2_w, 1), dtype=torch.float64, device=dev)        # cache1 (Q*X) and cache2 (Q'*X^T) are never live together
    staged_mine = torch.zeros(jpr * jobw if shard_rotcache else 1, dtype=torch.float64, device=dev)
    # one buffer for the gathered staging (Q*X) and the accumulators (Q'*X^T): never live together
    acc_w = 2 * colp if pipe_cols else ((nbr_x * D + (world * gpr - D)) * KP * outw if use_dist else 8)
    big = torch.zeros(max(world * jpr * jobw if shard_rotcache else 0, acc_w) * 8, dtype=torch.uint8, device=dev)
    staged_all = big.view(torch.float64)[: world * jpr * jobw] if shard_rotcache else None
    acc2 = big.view(torch.int64)[:acc_w]
    ctx.sync()

    phase_tot = {}

    def add_phases():
        for ph in PHASES:
            ms = ctx.phase_ms(ph)
            if ms >= 0:
                n = lib.sfg_last_phase_launches(ctx.h, ph.encode())
                a = phase_tot.setdefault(ph, [0.0, 0, 0.0])
                a[0] += ms
                a[1] += n
                a[2] += max(lib.sfg_last_phase_bytes(ctx.h, ph.encode()), 0.0)

    marks = []

    def mark(name):                                     # wall-time split of a step on the bench stream (solo-rank timing runs only)
        if solo:
            e = torch.cuda.Event(enable_timing=True)
            e.record(stream)
            marks.append((name, e))

    def step():
        mark("start")
        # (1) Q * X : output block columns of this rank
        if shard_rotcache:
            lib.sfg_ctx_clear_phases(ctx.h)
            chk(lib.sfg_rotcache_build_jobs_dev(ctx.h, C.c_void_p(A1.data_ptr()), KP, LEVEL, L, nbr_x, job0, job1, C.c_void_p(staged_mine.data_ptr())), "rotcache jobs")
            add_phases()
            mark("QX rotation-cache shard (key switching)")
            coll.all_gather(staged_all, staged_mine)
            chk(lib.sfg_rotcache_scatter_dev(ctx.h, C.c_void_p(staged_all.data_ptr()), KP, L, 0, njobs, 0, nbr_x, C.c_void_p(cache_buf.data_ptr())), "rotcache scatter")
            mark("QX scatter into the MAC layout (+ local copy standing in for the all-gather)")
            chk(lib.sfg_matmul_resident_range_rc_dev(ctx.h, C.c_void_p(cache_buf.data_ptr()), KP, L, gh, 0, 0, nblk_loc, C.c_void_p(out1.data_ptr())), "Q*X")
        else:
            chk(lib.sfg_matmul_resident_dev(ctx.h, C.c_void_p(A1.data_ptr()), KP, LEVEL, L, gh, 0, C.c_void_p(out1.data_ptr())), "Q*X")
        add_phases()
        mark("QX product (encode + MAC + giant-step alignment)")
        # (2) Q' * X^T : contraction over this rank's SNP blocks, combined before the giant steps
        lib.sfg_ctx_clear_phases(ctx.h)
        if not use_dist:                                # one rank: the one-shot product (the library's own accumulators; tests/test_gpu_properties.py: = accumulate + finalize)
            chk(lib.sfg_matmul_resident_dev(ctx.h, C.c_void_p(A2.data_ptr()), KP, LEVEL, L, gh, capi.SFG_TRANSPOSE, C.c_void_p(out2.data_ptr())), "Q'*X^T")
            add_phases()
            return
        if pipe_cols:
            chk(lib.sfg_rotcache_build_rows_dev(ctx.h, C.c_void_p(A2.data_ptr()), KP, LEVEL, L, nblk_loc, 0, nblk_loc, C.c_void_p(cache_buf.data_ptr())), "rotcache rows")
            add_phases()
            mark("QtXt rotation cache of the rank's block rows (key switching)")
            works = []
            for j in range(nbr_x):                      # column j is multiplied while column j-1 is reduce-scattered (sums < world * 2^46)
                buf = acc2[(j & 1) * colp: (j & 1) * colp + colp]
                if j >= 2:
                    works[j - 2].wait()
                lib.sfg_ctx_clear_phases(ctx.h)
                chk(lib.sfg_matmul_accumulate_rc_dev(ctx.h, C.c_void_p(cache_buf.data_ptr()), KP, L, gh, capi.SFG_TRANSPOSE,
                                                     0, nblk_loc, j, j + 1, 0, C.c_void_p(buf.data_ptr())), "Q'*X^T accumulate")
                add_phases()
                works.append(coll.reduce_scatter(acc_mine[j].view(-1), buf, async_op=True))
            for w in works[-2:]:
                w.wait()
        else:
            chk(lib.sfg_matmul_accumulate_dev(ctx.h, C.c_void_p(A2.data_ptr()), KP, LEVEL, L, gh, capi.SFG_TRANSPOSE,
                                              0, nblk_loc, 0, nbr_x, 0, C.c_void_p(acc2.data_ptr())), "Q'*X^T accumulate")
            add_phases()
            for j in range(nbr_x):                      # the window of the last giants runs into the next block column: those slots are ignored
                coll.reduce_scatter(acc_mine[j].view(-1), acc2[j * col: j * col + colp])
        mark("QtXt accumulate (encode + MAC; reduce-scatter stood in by a local copy)")
        lib.sfg_ctx_clear_phases(ctx.h)
        chk(lib.sfg_reduce_rows_dev(ctx.h, C.c_void_p(acc_mine.data_ptr()), nbr_x * gpr * KP * 2, L), "reduce acc")
        chk(lib.sfg_matmul_finalize_slots_dev(ctx.h, C.c_void_p(acc_mine.data_ptr()), KP, L, nbr_x, gpr, g_lo, 0, gpr, 0,
                                              C.c_void_p(out2.data_ptr())), "finalize")
        coll.all_reduce(out2)                           # aligned partial outputs of the ranks' giant shards
        chk(lib.sfg_reduce_rows_dev(ctx.h, C.c_void_p(out2.data_ptr()), KP * nbr_x * 2, L), "reduce out")
        add_phases()
        mark("QtXt reduce + giant-step alignment of the owned giants")

    def barrier():
        if use_dist and not solo:
            dist.barrier()
        torch.cuda.synchronize()

    # Plaintext coefficient cache (sfg_geno_set_plaintext_cache): enabled after the FIRST warm-up step, when every scratch pool has its final size, with what HBM
    # is then left beyond a reserve.  Blocks it holds skip skew + FFT in Q'*X^T of the same step and in both products of later steps (the reference keeps the
    # encoded diagonals across iterations too: MatMult4StreamPreprocess writes them to disk once, gwas/matmult.go:1228).  SFG_BENCH_PT_CACHE_GB=0 disables, a number caps.
    pt_cache = {"blocks": 0, "bytes": 0}

    def enable_pt_cache():
        want = os.environ.get("SFG_BENCH_PT_CACHE_GB", "auto")
        if want == "auto" and use_dist and args.backend == "gloo" and not solo:
            want = "0"                                  # rehearsal: the ranks share one GPU, "what is free" is not this rank's to take
        if want == "0":
            return
        torch.cuda.synchronize()
        free, _total = torch.cuda.mem_get_info(dev)
        budget = free - (16 << 30)
        if want != "auto":
            budget = min(budget, int(float(want) * (1 << 30)))
        if budget >= (1 << 29):
            chk(lib.sfg_geno_set_plaintext_cache(ctx.h, gh, C.c_size_t(budget)), "plaintext cache")
            pt_cache["budget"] = budget

    for w in range(args.warmup):
        step()
        if w == 0:
            enable_pt_cache()
    phase_tot.clear()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if use_dist and not solo:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        coll.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if "budget" in pt_cache:
        nb_, by_, hi_, fi_ = C.c_size_t(), C.c_size_t(), C.c_size_t(), C.c_size_t()
        chk(lib.sfg_geno_plaintext_cache_stats(ctx.h, gh, C.byref(nb_), C.byref(by_), C.byref(hi_), C.byref(fi_)), "plaintext cache stats")
        pt_cache.update(blocks=nb_.value, bytes=by_.value, hits=hi_.value, fills=fi_.value)
    near_ties = ctx.encoder_near_ties()                    # rounding audit of every encode since context creation (gate, warm-up, timed steps)
    unprovable = C.c_ulonglong()
    chk(lib.sfg_ctx_encoder_unprovable(ctx.h, C.byref(unprovable)), "encoder_unprovable")
    # ---- output digests, outside the timed region: SHA-256 over the per-ciphertext SHA-256s in global [i][j] order
    digests = None
    if not args.no_digest:
        def ct_hashes(t):                                   # t: [KP][ncols][outw] on the device -> bytes [KP][ncols][32]
            h = t.cpu().numpy()
            return np.frombuffer(b"".join(hashlib.sha256(h[i, j].tobytes()).digest() for i in range(h.shape[0]) for j in range(h.shape[1])),
                                 dtype=np.uint8).reshape(h.shape[0], h.shape[1], 32)
        mine = ct_hashes(out1)
        if use_dist and world > 1:
            parts = [None] * world
            dist.all_gather_object(parts, (blk0, mine))
            parts.sort(key=lambda p: p[0])
            mine = np.concatenate([p[1] for p in parts], axis=1)
        if rank == 0:
            digests = {"out1_sha256": hashlib.sha256(mine.tobytes()).hexdigest(),
                       "out2_sha256": hashlib.sha256(ct_hashes(out2).tobytes()).hexdigest(),
                       "of": "SHA-256 over the per-ciphertext SHA-256s in [i][j] order: Q*X (kp x m_ct) and Q'*X^T (kp x nbr); equal for every world size"}

    # ---- work accounting: useful ring-MACs = nrow*ncol*s*2 polys*L*(N/slots) per product
    macs_per_product = n_ind * m_snp * KP * 2 * L * (N // SLOTS)
    macs_per_step = 2 * macs_per_product
    value = macs_per_step * args.steps / dt
    res = {
        "metric": "pca_power_iter_ring_macs_per_s", "value": value, "unit": "ring-MAC/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "u64 ring words, exact integers: int8 digits on the matrix core for the MAC of all five moduli, fp64-held integers in the encode and key-switch kernels",
        "data": "synthetic",
        "config": {"workload": f"{args.config}: one PCA power iteration local work = Q*X + Q'*X^T, {n_ind} x {m_snp} int8 genotypes, "
                               f"kp={KP}, PN14QP438-shaped ring (N=16384, L=5 of 6 moduli), on-the-fly diagonal encode",
                   "parallelism": f"snp-block x{world}", "power_iter_wall_s": dt / args.steps,
                   "genotype_residency": "2-bit packed (sfg_geno_pack)" if args.packed_geno else "int8",
                   "engine": ("bench.py through torch.distributed (--engine torch: the A/B of the library's sfg_mgpu_* engine)" if use_dist else "single context"),
                   "collectives": (("RCCL" if args.backend == "nccl" else "gloo, host-staged (rehearsal: ranks may share a GPU; timing not meaningful)") if use_dist else "none"),
                   "plaintext_cache": (f"{pt_cache['blocks']} of {nblk_loc * nbr_x} blocks of this rank ({pt_cache['bytes'] / 2**30:.1f} GiB of HBM left after the first warm-up step; "
                                       f"{pt_cache.get('hits', 0)} block encodes served from it, {pt_cache.get('fills', 0)} filled)" if "budget" in pt_cache else "off"),
                   "rotation_cache_QX": ("sharded build + all-gather" if shard_rotcache else ("replicated" if use_dist else "single rank")),
                   "QtXt_reduce_scatter": ("per output block column, overlapped" if (use_dist and pipe_cols) else ("after the product" if use_dist else "none"))},
    }
    if solo:
        torch.cuda.synchronize()
        parts = {}
        per = len(marks) // (args.warmup + args.steps)
        for k in range(args.warmup * per, len(marks)):
            if marks[k][0] != "start":
                parts[marks[k][0]] = parts.get(marks[k][0], 0.0) + marks[k - 1][1].elapsed_time(marks[k][1]) / args.steps
        print(json.dumps({"solo_rank_timing_only": f"rank {rank} of {world}", "config": args.config, "ms_per_step": 1e3 * dt / args.steps,
                          "wall_ms_per_part": parts, "kernel_phases_ms_per_step": {k: v[0] / args.steps for k, v in phase_tot.items()},
                          "bytes": {"all_gather_recv_per_rank": (world - 1) * jpr * jobw * 8 if shard_rotcache else 0,
                                    "reduce_scatter_sent_per_rank": nbr_x * colp * 8 * (world - 1) // world,
                                    "all_reduce_out2": KP * nbr_x * outw * 8},
                          "plaintext_cache": pt_cache,
                          "note": "collectives replaced by local copies: the outputs are not a product"}), flush=True)
        lib.sfg_geno_free(ctx.h, gh)
        ctx.close()
        return
    if rank == 0:
        if gate is not None:
            res["parity_gate"] = gate
        if digests is not None:
            res["digests"] = digests
            res["digests_match_pinned"] = digests_match_pinned(args.config, digests)
        if world > 1:
            res["engine_fallback"] = bool(getattr(args, "engine_fallback", False))       # true: --allow-engine-fallback was given AND the library's engine could not be created
            res["rccl_ranks"] = dist.get_world_size() if use_dist and args.backend == "nccl" else 0
            res["rccl_ranks_of"] = "torch.distributed world size of the nccl (= RCCL) process group (--engine torch); 0 = gloo rehearsal"
        res["encoder_near_ties"] = {"count": near_ties, "within_2^-50": unprovable.value, "what": "encoder coefficients within 2^-40 of a rounding tie on rank 0 (0 = every plaintext provably "
                                                                 "rounded as the reference's 256-bit EncoderBig rounds it); a non-zero within_2^-50 count makes the library's synchronising entry points fail "
                                                                 "(about once per 10^15 coefficients; 2 x 10^11 per step here)"}
        # ---- roofline of the dominant kernel
        ms_small, n_small, by_small = phase_tot.get("mac_small", [0.0, 0, 0.0])
        if n_small and "mac_i8_pack_pt" in phase_tot:
            # default build: the small-modulus MAC runs on the int8 matrix core (mac_i8.hip) and is HBM bound; the kernel with the largest total time in the
            # timed region is then the panel NTT (fp64 vector issue bound).  Both are measured live (HIP events on the library's stream; the NTT on every 16th
            # of its ~50 000 launches per step, all launches counted) and the larger total is reported as `roofline`, the other one inside it.
            res["roofline"] = roofline_blocks(phase_tot, args, (n_ind, m_snp, nbr_x, mct_x, N, L, D, SLOTS, LEVEL), world, dt, value)
        elif n_small:
            nl_small = L - 1
            avg_ms = ms_small / n_small
            achieved = by_small / (ms_small * 1e-3) / 1e9
            padded_macs_s = 2 * nbr_x * mct_x * D * D * 2 * KP * nl_small * N * args.steps / world / (ms_small * 1e-3)   # this rank's share
            fma_s = 3.0 * padded_macs_s
            # SURVEY §8d algorithmic bytes of the whole step: int8 genotypes read once per product + ciphertexts in + out
            alg_step = 2 * n_ind * m_snp + (KP * nbr_x + KP * mct_x) * ctw * 8 + (KP * mct_x + KP * nbr_x) * outw * 8
            hbm_alg = alg_step * args.steps / dt / 1e9
            traffic, traffic_src = None, None
            try:        # HBM bytes per launch: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (tools/profile_pmc.sh), committed file
                for name in ("r03_pmc_traffic_per_launch.json", "r02_pmc_traffic_per_launch.json", "r01_pmc_traffic_per_launch.json"):
                    path = os.path.join(ROOT, "profiles", name)
                    if not os.path.exists(path):
                        continue
                    pm = json.load(open(path))
                    if pm.get("_config") == args.config:
                        ks = [k for k in pm if k.startswith("void k_mac_bc<false") or k.startswith("void k_mac_dma<false")]
                        best = max(ks, key=lambda k: pm[k]["launches"])
                        traffic, traffic_src = pm[best]["hbm_bytes_per_launch"], f"profiles/{name} (static: separate --pmc passes, not measured in this run)"
                        ref_alg = pm.get("_alg_bytes_per_launch_at_measurement")
                        if ref_alg:      # the passes ran at another launch size (SFG_MM_GROUP): same over-fetch ratio, this run's bytes per launch
                            traffic *= (by_small / n_small) / ref_alg
                            traffic_src += f"; measured at SFG_MM_GROUP={pm.get('_mm_group')} ({pm[best]['hbm_bytes_per_launch'] / ref_alg:.3f} x the algorithmic bytes) and scaled to this run's launch size"

                        break
            except Exception:
                pass
            res["roofline"] = {
                # the roofline that BINDS the dominant kernel (vector-ALU issue, not HBM): fp64 FMA rate of k_mac_bc, 2 flop per FMA,
                # against the fp64 vector peak of /opt/skills/guides/MI355X_MICROARCH.md (256 CUs x 4 SIMDs x 16 lanes x 2.4 GHz = 78.6 TFLOP/s)
                "bound": "valu_fp64", "achieved": 2.0 * fma_s / 1e12, "peak": 2.0 * FP64_VALU_SPEC_FMA_S / 1e12, "unit": "TFLOP/s",
                "frac": fma_s / FP64_VALU_SPEC_FMA_S,
                "traffic": traffic, "traffic_source": traffic_src,
                "kernel": "k_mac_bc<false, 30>", "avg_launch_ms": avg_ms, "launches": n_small,
                "what": "3 v_fmac_f64_dpp per padded ring-MAC (35-bit modulus: plaintext word split into 3 x 12-bit limbs, products < 2^48 summed "
                        "exactly; rot operand by row_newbcast); padded = the 91 x 91 diagonal slots of every block, 96-column tiles not counted",
                "fma_per_mac": 3, "padded_ring_macs_per_s_in_kernel": padded_macs_s, "fma_per_s_in_kernel": fma_s,
                "frac_of_measured_peak": fma_s / UBENCH_FMA_S, "measured_peak_fma_per_s": UBENCH_FMA_S, "measured_peak_source": UBENCH_FILE,
                "useful_fma_per_s_whole_step": 3.0 * value / world, "useful_frac_of_spec_whole_step": 3.0 * value / world / FP64_VALU_SPEC_FMA_S,
                # secondary: the HBM view the north star asks for
                "hbm_operands": {"achieved_GBps": achieved, "peak_GBps": HBM_PEAK_GBS, "frac": achieved / HBM_PEAK_GBS, "alg_bytes_per_launch": by_small / n_small,
                                 "what": "kernel operands of the same launches: fp64 rotation-cache slab + half-row plaintext panel + accumulator tile, each "
                                         "counted once (DESIGN.md §4); intermediates the encode / key-switch kernels wrote, not SURVEY §8(d)'s input/output bytes"},
                "hbm_algorithmic": {"bytes_per_step": alg_step, "achieved_GBps": hbm_alg, "frac": hbm_alg / HBM_PEAK_GBS,
                                    "what": "SURVEY §8(d): int8 genotypes once per product + ciphertexts in/out, divided by the WHOLE step time"}}
        res["phases_ms_per_step"] = {k: v[0] / args.steps for k, v in phase_tot.items()}
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(args.cpu_seconds, L, N, D)
            if not args.no_cpu_end_to_end:
                try:
                    res["cpu_baseline"]["end_to_end"] = cpu_end_to_end(ctx, P, res["cpu_baseline"]["cores"], res["cpu_baseline"]["cores_granted_vs_present"])
                except Exception as e:                       # noqa: BLE001 - a baseline leg must not lose the measured line
                    res["cpu_baseline"]["end_to_end"] = {"error": str(e)}
        print(json.dumps(res), flush=True)
    lib.sfg_geno_free(ctx.h, gh)
    ctx.close()
    if use_dist and not solo:
        dist.destroy_process_group()
    if gate is not None and gate["status"] != "ok":
        raise SystemExit("parity gate FAILED: the HIP path differs from the oracle")


if __name__ == "__main__":
    main()
