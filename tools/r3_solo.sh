# round 3: per-rank phase times of world sizes 2/4/8 at c4, one rank at a time on one GPU (collectives stood in by local copies; timing only)
cd $GRAFT_REPO_ROOT; TAG=${1:-r03_solo}; mkdir -p gpurun_out/$TAG
for rw in 0/8 7/8 0/4 0/2; do
  SFG_BENCH_SOLO=$rw timeout -k 10 400 python3 bench.py --gpus 1 --config c4 --steps 1 --warmup 1 > gpurun_out/$TAG/solo_${rw/\//of}.log 2>&1 || { tail -5 gpurun_out/$TAG/solo_${rw/\//of}.log; exit 1; }
  grep '^{' gpurun_out/$TAG/solo_${rw/\//of}.log >> gpurun_out/$TAG/solo_lines.jsonl
done
cat gpurun_out/$TAG/solo_lines.jsonl
