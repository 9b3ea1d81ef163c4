# round 4: streamed transposition (StagePack) - parity, then same-box A/B at c4 against the panel + transposition pass
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04stage
timeout -k 10 900 python -m pytest tests/test_gpu_matmul.py tests/test_gpu_properties.py tests/test_gpu_ptcache.py tests/test_gpu_fullsize.py -x -q -m gpu -k "not c4_100000" > gpurun_out/r04stage/tests.log 2>&1; rc=$?
tail -5 gpurun_out/r04stage/tests.log
[ $rc = 0 ] || exit $rc
for v in 1 0; do
  SFG_MAC_I8_STAGE=$v timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r04stage/bench_stage$v.json 2> gpurun_out/r04stage/bench_stage$v.err || { tail -5 gpurun_out/r04stage/bench_stage$v.err; exit 1; }
  python - <<P
import json
r=json.load(open("gpurun_out/r04stage/bench_stage$v.json"))
print("stage=$v", round(r["ms_per_step"]), {k:round(x) for k,x in r["phases_ms_per_step"].items()}, r["digests"]["out1_sha256"][:8], r["digests"]["out2_sha256"][:8], r["parity_gate"]["status"])
P
done
