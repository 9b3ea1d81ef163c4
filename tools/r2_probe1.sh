# round-2 probe 1: validate the packed-limb panel path, A/B it against the plain panel, collect SQ counters for the MAC kernels
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/p1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mac.py tests/test_gpu_matmul.py tests/test_gpu_properties.py "tests/test_gpu_fullsize.py::test_full_block_all_8192_diagonals_vs_oracle" "tests/test_gpu_fullsize.py::test_multi_group_two_pass_overlap_vs_oracle" -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for rep in 1 2; do for v in "SFG_X=0" "SFG_MAC_PT=plain"; do
env $v timeout 600 python bench.py --config c3 --no-cpu-baseline --no-check --no-digest 2>&1 | grep "^{" > /tmp/o.json
python -c "
import json; r=json.load(open('/tmp/o.json')); p=r['phases_ms_per_step']; print('%-20s total %.0f  encode %.0f  mac_small %.0f  mac_big %.0f  rotate %.0f skew %.0f' % ('$v', r['ms_per_step'], p['encode'], p['mac_small'], p['mac_big'], p['rotate'], p['skew']))"
done; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $GRAFT_REPO_ROOT/$O/counters.txt 2>&1
cd $GRAFT_REPO_ROOT
export SFG_MM_NO_OVERLAP=1 SFG_UPLOAD_BLOCKING=1
P1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
P2="SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"
i=0
for P in "$P1" "$P2"; do i=$((i+1))
  (cd /tmp && timeout 900 rocprofv3 --pmc $P --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc$i -o c -- python3 $GRAFT_REPO_ROOT/bench.py --config c2 --steps 1 --warmup 0 --no-cpu-baseline --no-check --no-digest > $GRAFT_REPO_ROOT/$O/pmc$i.log 2>&1)
  python3 - <<PY
import csv, glob, collections
fs = glob.glob("$O/pmc$i/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in fs:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
for k in agg:
    if "k_mac_dma" in k or "k_ntt_half3" in k or "k_fft_encode" in k:
        print(k, {c: "%.4g" % v for c, v in agg[k].items()})
PY
done
find $O -name "*counter_collection.csv" -size +20M -delete
