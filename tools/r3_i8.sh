# round 3: the experimental int8 matrix-core MAC (SFG_MAC_IMPL=i8): parity tests, then per-kernel times of one c2 bench pass against the default
cd $GRAFT_REPO_ROOT; TAG=${1:-r03_i8}; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
SFG_MAC_IMPL=i8 timeout -k 10 500 python3 -m pytest tests/test_gpu_matmul.py tests/test_gpu_fullsize.py -x -q -k "not c1_standin and not c5_batch" > gpurun_out/$TAG/test.log 2>&1 || { tail -30 gpurun_out/$TAG/test.log; exit 1; }
tail -2 gpurun_out/$TAG/test.log
for impl in bc i8; do
  SFG_MM_NO_OVERLAP=1 SFG_MAC_IMPL=$impl timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/p_$impl -o s -- python3 bench.py --config ${CFG:-c2} --no-cpu-baseline --no-check --steps 1 --warmup 1 ${PG:+--packed-geno} > gpurun_out/$TAG/bench_$impl.log 2>&1 || { tail -5 gpurun_out/$TAG/bench_$impl.log; exit 1; }
  cp $(find gpurun_out/$TAG/p_$impl -name '*kernel_stats.csv' | head -1) gpurun_out/$TAG/kernel_stats_$impl.csv; rm -rf gpurun_out/$TAG/p_$impl
  echo "== $impl"; grep '^{' gpurun_out/$TAG/bench_$impl.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['digests']['out1_sha256'][:12], d['digests']['out2_sha256'][:12])"
  head -8 gpurun_out/$TAG/kernel_stats_$impl.csv | cut -c1-50,140-260
done
