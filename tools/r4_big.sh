# round 4: changed subsystems under test, then the 46-bit modulus on the LDS-ring int8 MAC (SFG_MAC_I8_BIG=1) against the fp64 kernel, same box, c4
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04big
timeout -k 10 900 python -m pytest tests/test_gpu_diagcache.py tests/test_gpu_ptcache.py tests/test_refresh.py tests/test_gpu_pgen.py tests/test_gpu_encode.py tests/test_gpu_properties.py -x -q -m gpu > gpurun_out/r04big/tests.log 2>&1; rc=$?
tail -5 gpurun_out/r04big/tests.log
[ $rc = 0 ] || exit $rc
for v in 1 0; do
  SFG_MAC_I8_BIG=$v timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check > gpurun_out/r04big/bench_big$v.json 2> gpurun_out/r04big/bench_big$v.err || { tail -5 gpurun_out/r04big/bench_big$v.err; exit 1; }
  python - <<P
import json
r=json.load(open("gpurun_out/r04big/bench_big$v.json"))
print("big=$v", round(r["ms_per_step"]), {k:round(x) for k,x in r["phases_ms_per_step"].items()}, r["digests"]["out1_sha256"][:8], r["digests"]["out2_sha256"][:8])
P
done
