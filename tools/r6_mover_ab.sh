#!/bin/bash
# gpurun -- bash tools/r6_mover_ab.sh : (1) mover tuning, (2) 50k x 500k step with the pass (SFG_I8_MOVER=0) against the mover (default)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06mover; mkdir -p $O
timeout -k 10 600 python3 tools/r6_mover_tune.py 13 > $O/tune.txt 2>&1 || { tail -5 $O/tune.txt; exit 1; }
run() { local name=$1; shift
  env "$@" timeout -k 10 500 python3 bench.py --gpus 1 --config ${CFG:-c3} --steps 3 --warmup 2 --no-cpu-baseline --no-check > $O/$name.log 2>&1 || { tail -5 $O/$name.log; return 1; }
  python3 - "$O/$name.log" "$name" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
ph = d.get("phases_ms_per_step", {})
print(sys.argv[2], round(d["ms_per_step"]), d.get("digests", {}).get("out1_sha256", "")[:8], d.get("digests", {}).get("out2_sha256", "")[:8], {k: round(v, 1) for k, v in ph.items() if k in ("encode", "mac_i8_pack_pt", "mac_small", "mac_big")})
PY
}
run pass SFG_I8_MOVER=0 && run mover && run mover_d2 SFG_I8_MOVER_DEPTH=2 && run mover_2560 SFG_I8_MOVER=2560 && run pass_again SFG_I8_MOVER=0
