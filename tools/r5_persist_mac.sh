#!/bin/bash
# k_mac_i8_ringp (persistent workgroups, the LDS ring running through the pair boundaries) against k_mac_i8_ring: SFG_MAC_I8_PERSIST = workgroups per launch (0 = off)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05persist; mkdir -p $O
run() { local name=$1; shift
  env "$@" python3 bench.py --gpus 1 --config ${CFG:-c3} --steps 3 --warmup 2 --no-cpu-baseline --no-check > $O/$name.log 2>&1
  python3 - "$name" <<'PY'
import json, sys
d = json.loads([l for l in open(f"gpurun_out/r05persist/{sys.argv[1]}.log") if l.startswith("{")][-1])
ph = d.get("phases_ms_per_step", {})
print(sys.argv[1], round(d["ms_per_step"]), d.get("digests", {}).get("out1_sha256", "")[:8], d.get("digests", {}).get("out2_sha256", "")[:8], {k: round(v, 1) for k, v in ph.items() if k in ("mac_small", "mac_big", "mac_i8_pack_pt", "mac_i8_untile")})
PY
}
run off SFG_MAC_I8_PERSIST=0
for n in 256 512 1024 248; do run p$n SFG_MAC_I8_PERSIST=$n; done
run off_b SFG_MAC_I8_PERSIST=0
