cd $GRAFT_REPO_ROOT; TAG=${1:-r03_solo}; mkdir -p gpurun_out/$TAG
for rw in 0/8 7/8 0/2; do
  SFG_BENCH_ROTCACHE=replicated SFG_BENCH_SOLO=$rw timeout -k 10 400 python3 bench.py --gpus 1 --config c4 --steps 1 --warmup 1 > gpurun_out/$TAG/solo_rep_${rw/\//of}.log 2>&1 || { tail -5 gpurun_out/$TAG/solo_rep_${rw/\//of}.log; exit 1; }
  grep '^{' gpurun_out/$TAG/solo_rep_${rw/\//of}.log >> gpurun_out/$TAG/solo_replicated_lines.jsonl
done
cat gpurun_out/$TAG/solo_replicated_lines.jsonl
timeout -k 10 600 python3 bench.py --steps 2 --warmup 1 > gpurun_out/$TAG/bench_c4_n1.log 2>&1; tail -1 gpurun_out/$TAG/bench_c4_n1.log | cut -c1-900
