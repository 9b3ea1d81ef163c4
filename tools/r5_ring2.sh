#!/bin/bash
# k_mac_i8_ring2 (separate LDS rings for the rot and the plaintext tiles; SFG_MAC_I8_RING2=1: 3 + 3 slots, 2: 2 + 4 for the five-digit kernel) against k_mac_i8_ring
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05ring2; mkdir -p $O
run() { local name=$1; shift
  env "$@" python3 bench.py --gpus 1 --config ${CFG:-c3} --steps ${STEPS:-3} --warmup ${WARM:-2} --no-cpu-baseline --no-check > $O/$name.log 2>&1
  python3 - "$name" <<'PY'
import json, sys
d = json.loads([l for l in open(f"gpurun_out/r05ring2/{sys.argv[1]}.log") if l.startswith("{")][-1])
ph = d.get("phases_ms_per_step", {})
print(sys.argv[1], round(d["ms_per_step"]), d.get("digests", {}).get("out1_sha256", "")[:8], d.get("digests", {}).get("out2_sha256", "")[:8], {k: round(v, 1) for k, v in ph.items() if k in ("mac_small", "mac_big", "mac_i8_pack_pt")})
PY
}
run off
run r1 SFG_MAC_I8_RING2=1
run r2 SFG_MAC_I8_RING2=2
run off_b
run r1_b SFG_MAC_I8_RING2=1
