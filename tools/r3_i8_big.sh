# round 3: the 46-bit modulus on the fp64 DPP kernel vs on the int8 matrix core, c4, same box; parity tests first
cd $GRAFT_REPO_ROOT; TAG=${1:-r03_i8_big}; mkdir -p gpurun_out/$TAG
timeout -k 10 600 python3 -m pytest tests/test_gpu_matmul.py tests/test_gpu_fullsize.py tests/test_gpu_ptcache.py tests/test_gpu_diagcache.py -x -q -k "not c1_standin and not c5_batch" > gpurun_out/$TAG/test.log 2>&1 || { tail -30 gpurun_out/$TAG/test.log; exit 1; }
tail -1 gpurun_out/$TAG/test.log
for big in 0 1; do
  SFG_MAC_I8_BIG=$big SFG_BENCH_PT_CACHE_GB=0 timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-check --steps 2 --warmup 1 > gpurun_out/$TAG/bench_$big.log 2>&1 || { tail -5 gpurun_out/$TAG/bench_$big.log; exit 1; }
  grep '^{' gpurun_out/$TAG/bench_$big.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('big=$big', d['ms_per_step'], d['digests']['out1_sha256'][:12], d['digests']['out2_sha256'][:12], {k: round(v) for k, v in d['phases_ms_per_step'].items()})"
done
