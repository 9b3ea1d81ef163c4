#!/bin/bash
# streaming digit-plane stores in the plaintext NTT (now the default): encode batch size at 100k x 1M
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05ntstore; mkdir -p $O
run() { local name=$1; shift
  env "$@" python3 bench.py --gpus 1 --config c4 --steps 2 --warmup 1 --no-cpu-baseline --no-check > $O/$name.log 2>&1
  python3 - "$name" <<'PY'
import json, sys
d = json.loads([l for l in open(f"gpurun_out/r05ntstore/{sys.argv[1]}.log") if l.startswith("{")][-1])
ph = d.get("phases_ms_per_step", {})
print(sys.argv[1], round(d["ms_per_step"]), d.get("digests", {}).get("out1_sha256", "")[:8], d.get("digests", {}).get("out2_sha256", "")[:8], {k: round(v, 1) for k, v in ph.items() if k in ("encode", "mac_i8_pack_pt", "mac_small", "rotate")}, "ntt frac", round(d["roofline"]["frac"], 3), "us/launch", round(1e3 * d["roofline"]["avg_launch_ms"], 1))
PY
}
run c4_2048
run c4_1024 SFG_ENC_BATCH=1024
run c4_3072 SFG_ENC_BATCH=3072
run c4_2048_b
run c4_1536 SFG_ENC_BATCH=1536
