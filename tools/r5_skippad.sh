#!/bin/bash
# columns 91 .. 95 of the last plaintext tile neither written by the transposition nor fetched by the ring MAC (SFG_MAC_I8_SKIP_PAD=1)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05skippad; mkdir -p $O
run() { local name=$1; shift
  env "$@" python3 bench.py --gpus 1 --config ${CFG:-c3} --steps 3 --warmup 2 --no-cpu-baseline --no-check > $O/$name.log 2>&1
  python3 - "$name" <<'PY'
import json, sys
d = json.loads([l for l in open(f"gpurun_out/r05skippad/{sys.argv[1]}.log") if l.startswith("{")][-1])
ph = d.get("phases_ms_per_step", {})
print(sys.argv[1], round(d["ms_per_step"]), d.get("digests", {}).get("out1_sha256", "")[:8], d.get("digests", {}).get("out2_sha256", "")[:8], {k: round(v, 1) for k, v in ph.items() if k in ("mac_small", "mac_big", "mac_i8_pack_pt")})
PY
}
run off; run on SFG_MAC_I8_SKIP_PAD=1; run off_b; run on_b SFG_MAC_I8_SKIP_PAD=1; run off_c; run on_c SFG_MAC_I8_SKIP_PAD=1
