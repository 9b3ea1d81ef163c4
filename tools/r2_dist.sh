# the N > 1 code path (RCCL reduce-scatter over giants + all-reduce) at world size 1, digests vs the plain run
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${1:-dist1}; mkdir -p $O
timeout -k 10 400 python bench.py --config c3 --no-cpu-baseline 2>&1 | grep "^{" > $O/plain.json
SFG_BENCH_FORCE_COLLECTIVES=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --config c3 --no-cpu-baseline > $O/dist.log 2>&1
grep "^{" $O/dist.log > $O/dist.json
python - <<PY
import json
a=json.load(open("$O/plain.json")); b=json.load(open("$O/dist.json"))
print("plain", a["ms_per_step"], a["digests"]["out1_sha256"][:16], a["digests"]["out2_sha256"][:16], a["parity_gate"]["status"])
print("dist ", b["ms_per_step"], b["digests"]["out1_sha256"][:16], b["digests"]["out2_sha256"][:16], b["parity_gate"]["status"])
assert a["digests"]["out1_sha256"] == b["digests"]["out1_sha256"] and a["digests"]["out2_sha256"] == b["digests"]["out2_sha256"], "digests differ"
print("OK: collectives path gives identical digests")
PY
