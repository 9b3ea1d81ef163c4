"""bench.py's N > 1 path EXECUTED with world size > 1 on the one GPU a box has: `--backend gloo` stages the collectives through
host memory, so several ranks can share a device (RCCL refuses that).  Everything but the transport is the code the driver's
8-GPU run executes: per-rank SNP-block windows of one global matrix, per-rank ciphertext slices and seeds, the sharded
rotation-cache build + all-gather + scatter, the per-column reduce-scatter windows over the padded giant axis, finalize of the
owned giant slots, the all-reduce of the aligned partial outputs (pca.go:344,352).

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


def test_c2_two_ranks_share_the_gpu_with_a_sharded_rotation_cache_and_reproduce_the_single_process_digests(plain):
    """10 000 x 100 000 (13 SNP blocks -> 6 + 7; the last rank holds the ragged block).  Q*X's rotation cache built in shards (15 of the 30
    (block row, input) jobs per rank), all-gathered and scattered into the MAC layout; Q'*X^T one output block column at a time beside the previous
    column's reduce-scatter"""
    got = bench_line("c2", 2, "gloo", dict(SHARED_GPU_ENV, SFG_BENCH_ROTCACHE="sharded"))
    assert got["n_gpus"] == 2 and got["config"]["rotation_cache_QX"].startswith("sharded")
    assert got["config"]["QtXt_reduce_scatter"].startswith("per output block column")
    same(plain("c2"), got, "c2, 2 ranks, sharded cache")


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


def test_c3_two_ranks_share_the_gpu_and_reproduce_the_single_process_digests(plain):
    """50 000 x 500 000 (62 SNP blocks, 7 block rows of individuals): several MAC groups per rank, 7 reduce-scatter windows of which the last runs
    past its block column, 46 giant slots per rank of which the last rank owns 45; sharded rotation cache with a SHORT last shard (105 jobs = 53 + 52:
    the all-gather is padded)"""
    got = bench_line("c3", 2, "gloo", dict(SHARED_GPU_ENV, SFG_BENCH_ROTCACHE="sharded"))
    assert got["n_gpus"] == 2 and got["config"]["rotation_cache_QX"].startswith("sharded")
    same(plain("c3"), got, "c3, 2 ranks")


def test_collectives_path_at_world_size_1_over_rccl_matches_the_plain_path(plain):
    """the same sequence over RCCL (backend nccl) with one rank: stream ordering between the library's kernels and the collectives
    (bench.py once handed torch's default stream, handle 0 = "the context's own stream", to the library and the collectives read the accumulators early)"""
    got = bench_line("c3", 1, "nccl", None, force_coll=True)
    assert got["config"]["collectives"] == "RCCL"
    same(plain("c3"), got, "c3, forced collectives at world size 1")


def test_pipelined_and_sharded_sequences_over_rccl_at_world_size_1(plain):
    """the round-3 additions over the real transport: all_gather_into_tensor of the sharded rotation cache and the ASYNC per-column reduce-scatters
    (RCCL's own stream, work.wait() before a column buffer is reused) with one rank, 10 000 x 100 000"""
    got = bench_line("c2", 1, "nccl", {"SFG_BENCH_ROTCACHE": "sharded"}, force_coll=True)
    assert got["config"]["collectives"] == "RCCL" and got["config"]["rotation_cache_QX"].startswith("sharded")
    assert got["config"]["QtXt_reduce_scatter"].startswith("per output block column")
    same(plain("c2"), got, "c2, forced RCCL collectives, sharded cache, pipelined reduce-scatter")


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
