# SQ counters of the encode / NTT kernels (two PMC passes, c2, single queue): bash tools/r2_pmc_sq.sh <tag> [lib]
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${1:-pmcsq}; mkdir -p $O
export SFG_MM_NO_OVERLAP=1 SFG_UPLOAD_BLOCKING=1
[ -n "$2" ] && export SFG_LIB_PATH=$PWD/$2
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/p$i -o c -- python3 $GRAFT_REPO_ROOT/bench.py --config c2 --steps 1 --warmup 0 --no-cpu-baseline --no-check --no-digest > $GRAFT_REPO_ROOT/$O/p$i.log 2>&1) || { echo "pass $i failed"; tail -5 $O/p$i.log; exit 1; }
  python3 - <<PY | tee -a $O/sq_summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(int)
for f in glob.glob("$O/p$i/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
dur = collections.defaultdict(float)
for f in glob.glob("$O/p$i/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"][:40]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6; n[r["Kernel_Name"][:40]] += 1
for k in agg:
    if "k_ntt" in k or "k_fft" in k or "k_ksw" in k:
        print(k, "launches", n[k], "ms %.1f" % dur[k], {c: "%.4g" % v for c, v in sorted(agg[k].items())})
PY
done
find $O -name "*.csv" -size +2M -delete
