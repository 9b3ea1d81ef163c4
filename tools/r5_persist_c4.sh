#!/bin/bash
# the persistent ring MAC at 100k x 1M (k_mac_i8_ringp, SFG_MAC_I8_PERSIST=n workgroups; 0 = one workgroup per coefficient pair)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05persist; mkdir -p $O
run() { local name=$1; shift
  env "$@" python3 bench.py --gpus 1 --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-check > $O/$name.log 2>&1
  python3 - "$name" <<'PY'
import json, sys
d = json.loads([l for l in open(f"gpurun_out/r05persist/{sys.argv[1]}.log") if l.startswith("{")][-1])
ph = d.get("phases_ms_per_step", {})
print(sys.argv[1], round(d["ms_per_step"]), d.get("digests", {}).get("out1_sha256", "")[:8], d.get("digests", {}).get("out2_sha256", "")[:8], {k: round(v, 1) for k, v in ph.items() if k in ("mac_small", "mac_big", "mac_i8_pack_pt", "mac_i8_untile")})
PY
}
run c4_off SFG_MAC_I8_PERSIST=0
run c4_p256 SFG_MAC_I8_PERSIST=256
run c4_off_b SFG_MAC_I8_PERSIST=0
run c4_p256_b SFG_MAC_I8_PERSIST=256
