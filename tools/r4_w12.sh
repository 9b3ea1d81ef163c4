cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04w12
timeout -k 10 600 python -m pytest tests/test_gpu_mac.py tests/test_gpu_properties.py tests/test_gpu_matmul.py -x -q -m gpu > gpurun_out/r04w12/tests.log 2>&1; rc=$?
tail -4 gpurun_out/r04w12/tests.log
[ $rc = 0 ] || exit $rc
for v in 12 6 12; do
  SFG_MAC_I8_WAVES=$v timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check > gpurun_out/r04w12/bench_w$v.json 2> gpurun_out/r04w12/bench_w$v.err || { tail -5 gpurun_out/r04w12/bench_w$v.err; exit 1; }
  python - <<P
import json
r=json.load(open("gpurun_out/r04w12/bench_w$v.json"))
print("waves=$v", round(r["ms_per_step"]), {k:round(x) for k,x in r["phases_ms_per_step"].items()}, r["digests"]["out1_sha256"][:8], r["digests"]["out2_sha256"][:8])
P
done
