#!/bin/bash
# Final profiles of round 5.  (1) the driver's command unprofiled: the bench line with its live roofline and the CPU baseline (the oracle's native build is made
# first, so nothing compiles inside a timed or profiled run).  (2) the same command under rocprofv3 --kernel-trace --stats with --no-cpu-baseline (no make / gcc
# children under the profiler's preload): the per-kernel table.  Summaries only are kept.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r05final}; O=$R/gpurun_out/$TAG; mkdir -p $O; cd $R
make -C oracle native > $O/make.log 2>&1
( while sleep 60; do echo "tick $(date +%T)"; done ) & TICK=$!
timeout -k 10 700 python3 bench.py --steps ${STEPS:-20} --warmup ${WARMUP:-5} > $O/bench_default.log 2>&1 || { tail -5 $O/bench_default.log; kill $TICK; exit 1; }
grep '^{' $O/bench_default.log | tail -1 > $O/benchline.json
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-digest > $O/bench_prof.log 2>&1 || { tail -5 $O/bench_prof.log; kill $TICK; exit 1; }
kill $TICK
find $O -name "*kernel_trace.csv" -delete
python3 - "$TAG" <<'P'
import csv, glob, json, sys
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/{tag}/prof/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]:
    print(f"{r['Name'][:60]:60s} calls {int(r['Calls']):7d} total_ms {float(r['TotalDurationNs']) / 1e6:9.1f} avg_us {float(r['AverageNs']) / 1e3:10.1f}")
r = json.load(open(f"gpurun_out/{tag}/benchline.json"))
print("ms_per_step", round(r["ms_per_step"]), "roofline kernel", r["roofline"]["kernel"], r["roofline"]["avg_launch_ms"], "frac", round(r["roofline"]["frac"], 4),
      "| second", r["roofline"]["second_kernel"]["kernel"], r["roofline"]["second_kernel"]["avg_launch_ms"], "| cpu", r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"])
P
