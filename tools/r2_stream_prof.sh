# kernel times of the .bed streaming path (tools/bench_stream.py) under rocprofv3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-streamprof}; mkdir -p $R/gpurun_out/$TAG; cd $R
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/st -o s -- python3 tools/bench_stream.py > gpurun_out/$TAG/bench.log 2>&1
find gpurun_out/$TAG -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv, glob
for r in list(csv.DictReader(open(glob.glob("gpurun_out/$TAG/st/*kernel_stats.csv")[0])))[:14]:
    print(r["Name"][:44].ljust(44), r["Calls"].rjust(6), "total %.1f ms" % (float(r["TotalDurationNs"]) / 1e6), "avg %.3f ms" % (float(r["AverageNs"]) / 1e6))
PY
tail -2 gpurun_out/$TAG/bench.log | cut -c1-400
