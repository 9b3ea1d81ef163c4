# final profiles of round 4: the default bench command under rocprofv3 --kernel-trace --stats (one queue, the default), the same with the two-queue overlap, and the bench line
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04final; mkdir -p $O
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o p -- python3 $R/bench.py > $O/bench_default.log 2>&1 || { tail -5 $O/bench_default.log; exit 1; }
SFG_MM_OVERLAP=1 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_overlap -o p -- python3 $R/bench.py --no-cpu-baseline --no-check --no-digest > $O/bench_overlap.log 2>&1 || { tail -5 $O/bench_overlap.log; exit 1; }
cd $R
find $O -name "*kernel_trace.csv" -delete
grep '^{' $O/bench_default.log | tail -1 > $O/benchline.json
python3 - <<P
import csv,glob,json
for tag in ("default","overlap"):
    f=glob.glob("gpurun_out/r04final/prof_%s/**/*kernel_stats.csv"%tag, recursive=True)[0]
    print("==", tag)
    for r in list(csv.DictReader(open(f)))[:12]:
        print(f"{r['Name'][:60]:60s} calls {int(r['Calls']):7d} total_ms {float(r['TotalDurationNs'])/1e6:9.1f} avg_us {float(r['AverageNs'])/1e3:10.1f}")
r=json.load(open("gpurun_out/r04final/benchline.json"))
print(round(r["ms_per_step"]), r["roofline"]["avg_launch_ms"], r["roofline"]["second_kernel"]["avg_launch_ms"], r["cpu_baseline"]["value"], r["cpu_baseline"]["sample"][-160:])
P
