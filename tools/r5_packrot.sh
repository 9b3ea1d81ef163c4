#!/bin/bash
# k_i8_pack_rot: XCD-aware numbering of the workgroups that share a 128-byte source line (SFG_PACK_ROT_SPAN); 50k x 500k, ms per step and the phase's own time
# (the switch SFG_PACK_ROT_SPAN existed only for this measurement and was removed with the variant: DESIGN.md section 8, Round 5)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05packrot; mkdir -p $O
run() { local name=$1; shift
  env "$@" python3 bench.py --gpus 1 --config ${CFG:-c3} --steps 3 --warmup 2 --no-cpu-baseline --no-check > $O/$name.log 2>&1
  python3 - "$name" <<'PY'
import json, sys
d = json.loads([l for l in open(f"gpurun_out/r05packrot/{sys.argv[1]}.log") if l.startswith("{")][-1])
ph = d.get("phases_ms_per_step", {})
print(sys.argv[1], round(d["ms_per_step"]), d.get("digests", {}).get("out1_sha256", "")[:8], d.get("digests", {}).get("out2_sha256", "")[:8], {k: round(v, 1) for k, v in ph.items() if k in ("mac_i8_pack_rot", "mac_small", "mac_i8_pack_pt")})
PY
}
run span1 SFG_PACK_ROT_SPAN=1
run span2 SFG_PACK_ROT_SPAN=2
run span4 SFG_PACK_ROT_SPAN=4
run span1b SFG_PACK_ROT_SPAN=1
