# clock of the MAC kernels with and without their DMA traffic (diagnostic build): GRBM_GUI_ACTIVE / duration
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${1:-pmc2}; mkdir -p $O
export SFG_MM_NO_OVERLAP=1 SFG_UPLOAD_BLOCKING=1 SFG_LIB_PATH=$PWD/sfgwas_amd/lib_ab/libsfgwas_hip_d5.so
for dg in 0 3 8; do
  export SFG_MAC_DIAG=$dg
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/d$dg -o c -- python3 $GRAFT_REPO_ROOT/bench.py --config c2 --steps 1 --warmup 0 --no-cpu-baseline --no-check --no-digest > $GRAFT_REPO_ROOT/$O/d$dg.log 2>&1)
  python3 - <<PY | tee -a $O/clock_summary.txt
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob("$O/d$dg/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:40]][r["Counter_Name"]] += float(r["Counter_Value"])
dur = collections.defaultdict(float)
for f in glob.glob("$O/d$dg/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"][:40]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
for k in agg:
    if "k_mac" in k or "k_ntt_half3" in k or "k_fft" in k:
        print("diag $dg", k, "s %.4f" % dur[k], "clock GHz %.3f" % (agg[k]["GRBM_GUI_ACTIVE"] / 8 / dur[k] / 1e9), "valu wave-instr/s %.3e" % (agg[k]["SQ_INSTS_VALU"] / dur[k]))
PY
done
find $O -name "*.csv" -size +4M -delete
