#!/bin/bash
# Plaintext NTT with nontemporal digit-plane stores (-DSFG_NTT_NT_STORES, built as sfgwas_amd/lib/libsfgwas_hip_nt.so) against plain stores, at encode batches of 1024 / 2048 / 3072
# plaintexts per FFT / NTT launch pair: do streaming stores keep the FFT's coefficient rows cache resident at larger batches (fewer launches, each with ~ 10 us fixed cost)?
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05ntstore; mkdir -p $O
run() { local name=$1; shift
  env "$@" python3 bench.py --gpus 1 --config ${CFG:-c3} --steps 3 --warmup 2 --no-cpu-baseline --no-check > $O/$name.log 2>&1
  python3 - "$name" <<'PY'
import json, sys
d = json.loads([l for l in open(f"gpurun_out/r05ntstore/{sys.argv[1]}.log") if l.startswith("{")][-1])
ph = d.get("phases_ms_per_step", {})
print(sys.argv[1], round(d["ms_per_step"]), d.get("digests", {}).get("out1_sha256", "")[:8], {k: round(v, 1) for k, v in ph.items() if k in ("encode", "mac_i8_pack_pt", "mac_small")}, "ntt us/launch", round(1e3 * d["roofline"].get("avg_launch_ms", 0) if d["roofline"].get("kernel", "").startswith("k_ntt") else 1e3 * d["roofline"]["second_kernel"].get("avg_launch_ms", 0), 1))
PY
}
NT=SFG_LIB_PATH=$GRAFT_REPO_ROOT/sfgwas_amd/lib/libsfgwas_hip_nt.so
run plain_1024
run nt_1024 $NT
run plain_2048 SFG_ENC_BATCH=2048
run nt_2048 $NT SFG_ENC_BATCH=2048
run nt_3072 $NT SFG_ENC_BATCH=3072
run nt_4096 $NT SFG_ENC_BATCH=4096
