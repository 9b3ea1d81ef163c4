#!/bin/bash
# kernel-boundary cost of the encode launch pairs: plain kernel trace (no counters) of one c2 step, gaps between consecutive dispatches on the product's queue
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r05gaps}; CFG=${2:-c2}; mkdir -p $R/gpurun_out/$TAG; cd $R
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$TAG/t -o p -- python3 bench.py --config $CFG --steps 1 --warmup 1 --no-cpu-baseline --no-check --no-digest > gpurun_out/$TAG/run.log 2>&1
TAG=$TAG python3 - <<'PY' > gpurun_out/$TAG/summary.txt
import csv, glob, collections
rows = []
import os
for f in glob.glob("gpurun_out/" + os.environ["TAG"] + "/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0], r.get("Queue_Id", ""), r.get("Stream_Id", "")))
rows.sort()
rows = rows[len(rows) // 2:]          # the timed step (second half: after the warm-up step)
big = sorted(((b[0] - a[1], a[2][:40], b[2][:40]) for a, b in zip(rows, rows[1:])), reverse=True)[:12]
gap_after = collections.defaultdict(list); dur = collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    dur[a[2]].append(a[1] - a[0])
    gap_after[(a[2][:28], b[2][:28])].append(b[0] - a[1])
tot = rows[-1][1] - rows[0][0]
busy = sum(r[1] - r[0] for r in rows)
print(f"span {tot / 1e6:.1f} ms, sum of kernel durations {busy / 1e6:.1f} ms, {len(rows)} dispatches, mean gap {(tot - busy) / max(len(rows) - 1, 1) / 1e3:.2f} us")
for k, v in sorted(gap_after.items(), key=lambda kv: -sum(kv[1]))[:14]:
    v2 = sorted(v)
    print(f"{k[0]:28s} -> {k[1]:28s} n={len(v):6d} gap mean {sum(v) / len(v) / 1e3:7.2f} us  median {v2[len(v2) // 2] / 1e3:7.2f} us  total {sum(v) / 1e6:8.2f} ms")
print("largest single gaps (us):", [(round(g / 1e3), x, y) for g, x, y in big])
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:12]:
    print(f"{k[:44]:44s} n={len(v):6d} dur mean {sum(v) / len(v) / 1e3:8.2f} us total {sum(v) / 1e6:8.2f} ms")
PY
cat gpurun_out/$TAG/summary.txt
