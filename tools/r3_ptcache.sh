# round 3: the plaintext coefficient cache, measured: rank 0 of 8 at c4 (solo timing) with the cache off / auto, then (FULL=1) the one-GPU bench line with digests
cd $GRAFT_REPO_ROOT; TAG=${1:-r03_ptcache}; mkdir -p gpurun_out/$TAG; rm -f gpurun_out/$TAG/solo_lines.jsonl
timeout -k 10 300 python3 -m pytest tests/test_gpu_ptcache.py -x -q > gpurun_out/$TAG/test.log 2>&1 || { tail -30 gpurun_out/$TAG/test.log; exit 1; }
tail -1 gpurun_out/$TAG/test.log
for gb in ${MODES:-0 auto}; do
  SFG_BENCH_PT_CACHE_GB=$gb SFG_BENCH_SOLO=0/8 timeout -k 10 300 python3 bench.py --gpus 1 --config c4 --steps 2 --warmup 2 > gpurun_out/$TAG/solo_0of8_$gb.log 2>&1 || { tail -5 gpurun_out/$TAG/solo_0of8_$gb.log; exit 1; }
  grep '^{' gpurun_out/$TAG/solo_0of8_$gb.log >> gpurun_out/$TAG/solo_lines.jsonl
  echo "solo $gb done"
done
if [ -n "$FULL" ]; then
  timeout -k 10 500 python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline > gpurun_out/$TAG/bench_n1.log 2>&1 || { tail -5 gpurun_out/$TAG/bench_n1.log; exit 1; }
  grep '^{' gpurun_out/$TAG/bench_n1.log > gpurun_out/$TAG/bench_n1.json
fi
cat gpurun_out/$TAG/solo_lines.jsonl
