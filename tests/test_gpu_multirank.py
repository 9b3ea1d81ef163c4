"""bench.py's N > 1 paths EXECUTED with world size > 1 on the one GPU a box has.

(1) --engine torch (the A/B baseline since round 5: the sequence issued from bench.py through torch.distributed): `--backend gloo` stages the collectives through
host memory, so several ranks can share a device (RCCL refuses that).  Everything but the transport is the code the driver's
8-GPU run executes: per-rank SNP-block windows of one global matrix, per-rank ciphertext slices and seeds, the sharded
rotation-cache build + all-gather + scatter, the per-column reduce-scatter windows over the padded giant axis, finalize of the
owned giant slots, the all-reduce of the aligned partial outputs (pca.go:344,352).

(2) --engine lib (the default for N > 1 since round 5): the sequence runs INSIDE libsfgwas_hip (sfg_mgpu_*, mgpu.hip) - as one process driving N ranks (the Go
party's form; ranks on one device use the library's in-process direct transport) and under the driver's one-process-per-GPU launcher (RCCL, world 1 here).

The bar: the SHA-256 digests of both products printed by every world size equal the single-process line's - every output
word of Q*X and Q'*X^T is the same whatever the sharding."""
import json
import os
import subprocess
import sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--no-cpu-baseline", "--no-check", "--warmup", "0", "--steps", "1"]
# ranks that share one GPU also share its 288 GB: smaller MAC groups / accumulator passes (results do not depend on either:
# tests/test_gpu_properties.py), and Q'*X^T's rank-local rotation cache only when it is small
SHARED_GPU_ENV = {"SFG_MM_GROUP": "4", "SFG_MM_ACC_BUDGET_MB": "4096", "SFG_BENCH_CACHE2_GB": "24"}
_port = [29600]


def bench_line(config, world=1, backend=None, env=None, force_coll=False):
    e = dict(os.environ)
    e.update(env or {})
    args = ["bench.py", "--config", config, "--gpus", str(world)] + COMMON
    if world == 1 and not force_coll:
        cmd = [sys.executable] + args
    else:
        _port[0] += 1
        if force_coll:
            e["SFG_BENCH_FORCE_COLLECTIVES"] = "1"
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(_port[0])] + args + (["--backend", backend] if backend else [])
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


@pytest.fixture(scope="module")
def plain():
    cache = {}

    def get(config):
        if config not in cache:
            cache[config] = bench_line(config)
            assert cache[config]["n_gpus"] == 1 and cache[config]["config"]["collectives"] == "none"
        return cache[config]
    return get


def same(a, b, what):
    assert a["digests"]["out1_sha256"] == b["digests"]["out1_sha256"], f"{what}: Q*X differs from the single-process product"
    assert a["digests"]["out2_sha256"] == b["digests"]["out2_sha256"], f"{what}: Q'*X^T differs from the single-process product"


def test_c2_three_ranks_share_the_gpu_and_reproduce_the_single_process_digests(plain):
    """4 + 4 + 5 SNP blocks; 31 giant slots per rank (93 > 91: the last rank owns 29); the default multi-GPU configuration"""
    got = bench_line("c2", 3, "gloo", SHARED_GPU_ENV)
    assert got["n_gpus"] == 3 and got["config"]["rotation_cache_QX"] == "replicated"
    assert got["config"]["QtXt_reduce_scatter"].startswith("per output block column")
    same(plain("c2"), got, "c2, 3 ranks")


def test_c2_three_ranks_sharded_cache_and_unpipelined_reduce_scatter(plain):
    """the other A/B switch: reduce-scatters issued after the whole Q'*X^T accumulate (the path a rank takes when its own rotation cache does not
    fit), whose last window runs past its block column into the padding"""
    env = dict(SHARED_GPU_ENV, SFG_BENCH_ROTCACHE="sharded", SFG_BENCH_CACHE2_GB="0")
    got = bench_line("c2", 3, "gloo", env)
    assert got["config"]["rotation_cache_QX"].startswith("sharded") and got["config"]["QtXt_reduce_scatter"] == "after the product"
    same(plain("c2"), got, "c2, 3 ranks, sharded cache, unpipelined")


def test_collectives_path_at_world_size_1_over_rccl_matches_the_plain_path(plain):
    """the same sequence over RCCL (backend nccl) with one rank: stream ordering between the library's kernels and the collectives
    (bench.py once handed torch's default stream, handle 0 = "the context's own stream", to the library and the collectives read the accumulators early)"""
    got = bench_line("c2", 1, "nccl", None, force_coll=True)
    assert got["config"]["collectives"] == "RCCL"
    same(plain("c2"), got, "c2, forced collectives at world size 1")


def test_bench_launches_its_own_ranks_when_asked_for_more_than_one_gpu(plain):
    """`python bench.py --gpus 2` with no launcher around it (no RANK in the environment) starts its two ranks itself - child processes under
    torch.distributed.run, before the parent touches the GPU - and relays rank 0's line.  Same digests as the single process."""
    e = dict(os.environ)
    e.update(SHARED_GPU_ENV)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    r = subprocess.run([sys.executable, "bench.py", "--config", "c2", "--gpus", "2", "--backend", "gloo"] + COMMON, cwd=ROOT, env=e, capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["collectives"].startswith("gloo")
    same(line, plain("c2"), "self-launched 2 ranks")


# ---- the same bar for the multi-GPU sequence INSIDE the library (bench.py --engine lib, the default for N > 1: sfg_mgpu_*, sfgwas_amd/csrc/mgpu.hip)
LIB_SHARED_ENV = {"SFG_MM_GROUP": "4", "SFG_MM_ACC_BUDGET_MB": "4096", "SFG_MGPU_CACHE_GB": "24"}


def lib_line(config, world, extra=(), env=None, launcher=False):
    e = dict(os.environ)
    e.update(env or {})
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        e.pop(k, None)
    args = ["bench.py", "--config", config, "--gpus", str(world)] + COMMON + list(extra)
    if launcher:
        _port[0] += 1
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1", "--master-port", str(_port[0])] + args
    else:
        cmd = [sys.executable] + args
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_library_engine_one_process_two_ranks_on_one_device_reproduces_the_single_gpu_digests(plain):
    """the Go party's form: ONE process, sfg_mgpu_create(devices = [0, 0]) - a repeated device, so the in-process direct transport carries the reduce-scatters and
    the all-reduce; 10 000 x 100 000 = 13 SNP blocks -> 6 + 7"""
    got = lib_line("c2", 2, ["--single-process", "--devices", "0,0"], LIB_SHARED_ENV)
    assert got["n_gpus"] == 2 and got["config"]["engine"].startswith("libsfgwas_hip sfg_mgpu") and got["config"]["collectives"].startswith("in-process direct")
    same(plain("c2"), got, "c2, library engine, 2 ranks in one process")


def test_library_engine_two_ranks_c3_and_three_ranks_unpipelined_c2(plain):
    """50 000 x 500 000 on two ranks (62 SNP blocks -> 31 + 31, 7 output block columns: the two column buffers change hands five times; 46 giant slots per rank,
    92 > 91) against the digests bench.py pins for c3 (the line says so itself: digests_match_pinned), and the unpipelined form at 10 000 x 100 000 on three ranks.
    The line of an N > 1 run states what it ran on: the engine, no fallback, the pre-flight, the rank count the communicator reports (0: direct transport)."""
    got = lib_line("c3", 2, ["--single-process", "--devices", "0,0"], LIB_SHARED_ENV)
    assert got["digests_match_pinned"] is True, got["digests"]
    assert got["engine_fallback"] is False and got["rccl_ranks"] == 0 and got["preflight"].startswith("ok")
    got = lib_line("c2", 3, ["--single-process", "--devices", "0,0,0"], dict(LIB_SHARED_ENV, SFG_MGPU_CACHE_GB="0"))
    same(plain("c2"), got, "c2, library engine, 3 ranks, reduce-scatters after the product")


def test_library_engine_over_rccl_at_world_1_in_both_process_models(plain):
    """the library's own RCCL calls (ncclReduceScatter on the collectives' queue beside the next column's product, ncclAllReduce) with one rank: as one process
    (ncclCommInitAll) and under the launcher the driver uses (one process per GPU: sfg_mgpu_unique_id on rank 0 -> TCP store -> sfg_mgpu_create_rank)"""
    got = lib_line("c2", 1, [], {"SFG_MGPU_FORCE_COLLECTIVES": "1"})
    assert got["config"]["collectives"].startswith("RCCL (called by the library")
    assert got["rccl_ranks"] == 1 and got["engine_fallback"] is False and got["preflight"].startswith("ok")       # ncclCommCount of the one-rank communicator
    same(plain("c2"), got, "c2, library engine, RCCL at world 1, one process")
    got = lib_line("c2", 1, [], {"SFG_MGPU_FORCE_COLLECTIVES": "1"}, launcher=True)
    assert got["config"]["engine"].endswith("sfg_mgpu_create_rank") and got["config"]["collectives"].startswith("RCCL")
    same(plain("c2"), got, "c2, library engine, RCCL at world 1, launcher")
