# per-kernel times per step of rank 0 of 8 at c4 (solo timing) with the plaintext cache off / auto
cd $GRAFT_REPO_ROOT; TAG=${1:-r03_ptcache_prof}; mkdir -p gpurun_out/$TAG; export TMPDIR=/tmp
for gb in ${MODES:-0 auto}; do
  SFG_BENCH_PT_CACHE_GB=$gb SFG_BENCH_SOLO=0/8 timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$TAG/prof_$gb -o p -- python3 bench.py --gpus 1 --config c4 --steps 1 --warmup 2 > gpurun_out/$TAG/prof_$gb.log 2>&1 || { tail -5 gpurun_out/$TAG/prof_$gb.log; exit 1; }
  f=$(find gpurun_out/$TAG/prof_$gb -name '*kernel_trace.csv' | head -1)
  python3 tools/trace_steps.py $f 28 gpurun_out/$TAG/seq_$gb.txt > gpurun_out/$TAG/steps_$gb.txt
  rm -rf gpurun_out/$TAG/prof_$gb
  echo "== $gb"; grep -A7 "== step 2" gpurun_out/$TAG/steps_$gb.txt
done
