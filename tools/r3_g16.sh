# round 3: c4 with 2-bit packed residency, MAC groups of 8 vs 16 block rows (same box, back to back)
cd $GRAFT_REPO_ROOT; TAG=${1:-r03_g16}; mkdir -p gpurun_out/$TAG
for g in 8 16 8 16; do
  SFG_MM_GROUP=$g timeout -k 10 400 python3 bench.py --packed-geno --steps 2 --warmup 1 --no-cpu-baseline --no-check > gpurun_out/$TAG/g$g.log 2>&1 || { tail -5 gpurun_out/$TAG/g$g.log; exit 1; }
  python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/$TAG/g$g.log") if l.startswith("{")][-1])
print("group", $g, "ms_per_step", round(d["ms_per_step"],1), "digests", d["digests"]["out1_sha256"][:12], d["digests"]["out2_sha256"][:12], {k: round(v) for k,v in d["phases_ms_per_step"].items()})
PY
done | tee gpurun_out/$TAG/summary.txt
