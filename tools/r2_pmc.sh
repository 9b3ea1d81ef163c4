# SQ / clock counters of the MAC kernels (c2, single queue, blocking uploads: see tools/profile_pmc.sh)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${1:-pmc1}; mkdir -p $O
export SFG_MM_NO_OVERLAP=1 SFG_UPLOAD_BLOCKING=1
P1="GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY"
P2="SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM"
P3="TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"
i=0
for P in "$P1" "$P2" "$P3"; do i=$((i+1))
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 600 rocprofv3 --pmc $P --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc$i -o c -- python3 $GRAFT_REPO_ROOT/bench.py --config ${2:-c2} --steps 1 --warmup 0 --no-cpu-baseline --no-check --no-digest > $GRAFT_REPO_ROOT/$O/pmc$i.log 2>&1)
  python3 - <<PY | tee -a $O/pmc_summary.txt
import csv, glob, collections
fs = glob.glob("$O/pmc$i/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(set)
for f in fs:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:48]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k].add(r["Dispatch_Id"])
dur = collections.defaultdict(float)
for f in glob.glob("$O/pmc$i/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Kernel_Name"][:48]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
for k in agg:
    if "k_mac" in k or "k_ntt_half3" in k or "k_fft_encode" in k:
        print(k, "launches", len(n[k]), "ms %.1f" % dur.get(k, 0), {c: "%.4g" % v for c, v in agg[k].items()})
PY
done
find $O -name "*counter_collection.csv" -size +8M -delete; find $O -name "*kernel_trace.csv" -size +8M -delete
