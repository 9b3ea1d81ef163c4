# kernel table of the default build on ONE queue (every kernel alone): the honest per-kernel times
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; mkdir -p $R/gpurun_out/r04single
export SFG_MM_NO_OVERLAP=1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r04single/prof -o p -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-check --no-digest > $R/gpurun_out/r04single/prof.log 2>&1
cd $R
find gpurun_out/r04single/prof -name "*kernel_trace.csv" -delete
python3 - <<P
import csv,glob
f=glob.glob("gpurun_out/r04single/prof/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:24]:
    print(f"{r['Name'][:70]:70s} calls {int(r['Calls']):7d} total_ms {float(r['TotalDurationNs'])/1e6:9.1f} avg_us {float(r['AverageNs'])/1e3:10.1f} {r['Percentage']}%")
P
tail -2 gpurun_out/r04single/prof.log | cut -c1-300
