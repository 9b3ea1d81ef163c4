cd $GRAFT_REPO_ROOT; TAG=${1:-r03_auto}; mkdir -p gpurun_out/$TAG
for mode in auto g8; do
  if [ $mode = g8 ]; then export SFG_MM_GROUP=8; else unset SFG_MM_GROUP; fi
  timeout -k 10 500 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/$TAG/$mode.log 2>&1 || { tail -8 gpurun_out/$TAG/$mode.log; exit 1; }
  python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/$TAG/$mode.log") if l.startswith("{")][-1])
print("$mode", "ms_per_step", round(d["ms_per_step"],1), "gate", d["parity_gate"]["status"], "digests", d["digests"]["out1_sha256"][:12], d["digests"]["out2_sha256"][:12], {k: round(v) for k,v in d["phases_ms_per_step"].items()}, "mac launches", d["roofline"]["launches"])
PY
done | tee gpurun_out/$TAG/summary.txt
rocm-smi --showmeminfo vram 2>/dev/null | tail -3
