# round 4: the LDS-ring int8 MAC - parity first, then same-box A/B at c4 against the cache-shared kernel
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04ring
timeout -k 10 600 python -m pytest tests/test_gpu_mac.py tests/test_gpu_properties.py tests/test_gpu_matmul.py -x -q -m gpu > gpurun_out/r04ring/tests.log 2>&1; rc=$?
tail -5 gpurun_out/r04ring/tests.log
[ $rc = 0 ] || exit $rc
for v in ring cache ring; do
  SFG_MAC_I8_ROT=$v timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04ring/bench_$v.json 2> gpurun_out/r04ring/bench_$v.err || { tail -5 gpurun_out/r04ring/bench_$v.err; exit 1; }
  python - <<P
import json
r=json.load(open("gpurun_out/r04ring/bench_$v.json"))
print("$v", round(r["ms_per_step"]), {k:round(x) for k,x in r["phases_ms_per_step"].items()}, r["digests"]["out1_sha256"][:8], r["digests"]["out2_sha256"][:8], r["parity_gate"]["status"])
P
done
