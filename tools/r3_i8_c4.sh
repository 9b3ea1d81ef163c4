# round 3: default vs experimental int8 MAC at c4 on one box (two queues as in production), digests compared
cd $GRAFT_REPO_ROOT; TAG=${1:-r03_i8_c4}; mkdir -p gpurun_out/$TAG
for impl in bc i8; do
  SFG_MAC_IMPL=$impl SFG_BENCH_PT_CACHE_GB=0 timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-check --steps 2 --warmup 1 ${PG:+--packed-geno} > gpurun_out/$TAG/bench_$impl.log 2>&1 || { tail -5 gpurun_out/$TAG/bench_$impl.log; exit 1; }
  grep '^{' gpurun_out/$TAG/bench_$impl.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$impl', d['ms_per_step'], d['digests']['out1_sha256'][:12], d['digests']['out2_sha256'][:12], {k: round(v) for k, v in d['phases_ms_per_step'].items()})"
done
