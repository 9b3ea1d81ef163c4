# how much wall time does key switching cost?  single queue (kernel times undisturbed) against the default two-queue overlap, then the kernel table of the default
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04ovl
for v in 1 0; do
  if [ $v = 1 ]; then export SFG_MM_NO_OVERLAP=1; else unset SFG_MM_NO_OVERLAP; fi
  timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-digest > gpurun_out/r04ovl/bench_noovl$v.json 2> gpurun_out/r04ovl/bench_noovl$v.err || { tail -5 gpurun_out/r04ovl/bench_noovl$v.err; exit 1; }
  python - <<P
import json
r=json.load(open("gpurun_out/r04ovl/bench_noovl$v.json"))
print("no_overlap=$v", round(r["ms_per_step"]), {k:round(x) for k,x in r["phases_ms_per_step"].items()})
P
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04ovl/prof -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-check --no-digest > $GRAFT_REPO_ROOT/gpurun_out/r04ovl/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r04ovl/prof -name "*kernel_trace.csv" -delete
head -30 $(find gpurun_out/r04ovl/prof -name "*kernel_stats.csv" | head -1) | cut -c1-160
