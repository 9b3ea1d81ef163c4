# round 3: int8 MAC with the rot tiles through the cache vs staged through LDS, c4, same box; parity tests first
cd $GRAFT_REPO_ROOT; TAG=${1:-r03_i8_lds}; mkdir -p gpurun_out/$TAG
timeout -k 10 500 python3 -m pytest tests/test_gpu_matmul.py tests/test_gpu_fullsize.py tests/test_gpu_ptcache.py -x -q -k "not c1_standin and not c5_batch" > gpurun_out/$TAG/test.log 2>&1 || { tail -30 gpurun_out/$TAG/test.log; exit 1; }
tail -1 gpurun_out/$TAG/test.log
for rot in cache lds; do
  SFG_MAC_I8_ROT=$rot SFG_BENCH_PT_CACHE_GB=0 timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-check --steps 2 --warmup 1 > gpurun_out/$TAG/bench_$rot.log 2>&1 || { tail -5 gpurun_out/$TAG/bench_$rot.log; exit 1; }
  grep '^{' gpurun_out/$TAG/bench_$rot.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$rot', d['ms_per_step'], d['digests']['out1_sha256'][:12], d['digests']['out2_sha256'][:12], {k: round(v) for k, v in d['phases_ms_per_step'].items()})"
done
