#!/bin/bash
# fixed cost per launch of the encode kernels: kernel durations (plain kernel trace) against the number of plaintexts per launch pair (SFG_ENC_BATCH)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r05encb}; CFG=${2:-c2}; mkdir -p $R/gpurun_out/$TAG; cd $R
for b in 205 410 1024 2048 4096 8192; do
  SFG_ENC_BATCH=$b SFG_BENCH_PT_CACHE_GB=0 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$TAG/b$b -o p -- python3 bench.py --config $CFG --steps 1 --warmup 0 --no-cpu-baseline --no-check --no-digest > gpurun_out/$TAG/b$b.log 2>&1
done
python3 - <<'PY' > gpurun_out/r05encb/summary.txt
import csv, glob, collections
for b in (205, 410, 1024, 2048, 4096, 8192):
    dur = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/r05encb/b{b}/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"].split("(")[0]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k in ("void k_ntt_half3<false, true>", "void k_fft_encode<false>"):
        v = dur.get(k, [])
        if v:
            full = [x for x in v if x > 0.7 * max(v)]      # launches of a whole batch (a block's last batch is short)
            print(f"batch {b:5d} {k:32s} launches {len(v):5d} total {sum(v) / 1e6:8.2f} ms  full-batch launch {sum(full) / len(full) / 1e3:8.2f} us = {sum(full) / len(full) / b:7.2f} ns per plaintext")
PY
cat gpurun_out/$TAG/summary.txt
find gpurun_out/$TAG -name "*.csv" -size +5M -delete
