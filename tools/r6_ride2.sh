#!/bin/bash
# gpurun -- bash tools/r6_ride2.sh : mover shapes of the riding transposition at 50k x 500k (A/B build): "nblocks depth nt"
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06ride2; mkdir -p $O
AB=$GRAFT_REPO_ROOT/sfgwas_amd/lib_ab/libsfgwas_hip.so
run() { local name=$1; shift
  env SFG_LIB_PATH=$AB "$@" timeout -k 10 500 python3 bench.py --gpus 1 --config ${CFG:-c3} --steps ${STEPS:-3} --warmup 2 --no-cpu-baseline --no-check --no-digest > $O/$name.log 2>&1 || { tail -5 $O/$name.log; return 1; }
  python3 - "$O/$name.log" "$name" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
ph = d.get("phases_ms_per_step", {})
r = d.get("roofline", {})
print(sys.argv[2], round(d["ms_per_step"]), {k: round(v, 1) for k, v in ph.items() if k in ("encode", "mac_i8_pack_pt", "mac_small", "mac_big")}, r.get("kernel"), round(r.get("avg_launch_ms", 0), 4), r.get("launches"), "mac launches", r.get("second_kernel", {}).get("launches"))
PY
}
for v in ${VARIANTS:-"192 1 1" "128 3 1" "96 3 1" "160 2 1" "192 2 1" "192 1 0" "64 3 1"}; do set -- $v; run ride_$1_$2_$3 SFG_PT_RIDE=$1 SFG_PT_RIDE_DEPTH=$2 SFG_PT_RIDE_NT=$3 || exit 1; done
