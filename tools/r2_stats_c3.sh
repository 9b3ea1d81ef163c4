cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-c3stats}; mkdir -p $R/gpurun_out/$TAG; cd $R
export SFG_MM_NO_OVERLAP=1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/stats -o stats -- python3 bench.py --config c3 --no-cpu-baseline --no-check --no-digest > gpurun_out/$TAG/bench.log 2>&1
find gpurun_out/$TAG -name "*kernel_trace.csv" -delete; find gpurun_out/$TAG -name "*.db" -delete
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/$TAG/stats/stats_kernel_stats.csv")))
for r in rows[:18]:
    print(f"{r['Name'][:50]:50s} calls {int(r['Calls']):6d} total {float(r['TotalDurationNs'])/2e6:8.1f} ms/pass avg {float(r['AverageNs'])/1e3:9.1f} us")
PY
