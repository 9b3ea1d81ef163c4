# sketch kernel time under rocprofv3 (100k x 400k): bash tools/r2_sketch.sh <tag> [lib]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-sketch2}; mkdir -p $R/gpurun_out/$TAG; cd $R
[ -n "$2" ] && export SFG_LIB_PATH=$R/$2
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/st -o s -- python3 tools/bench_sketch.py > gpurun_out/$TAG/bench.log 2>&1
find gpurun_out/$TAG -name "*kernel_trace.csv" -delete
python3 - <<PY
import csv, glob
for r in csv.DictReader(open(glob.glob("gpurun_out/$TAG/st/*kernel_stats.csv")[0])):
    if "sketch" in r["Name"] or "colsums" in r["Name"]: print(r["Name"][:40], r["Calls"], "avg %.2f ms" % (float(r["AverageNs"]) / 1e6), "max %.2f ms" % (float(r["MaxNs"]) / 1e6))
PY
tail -2 gpurun_out/$TAG/bench.log | cut -c1-250
