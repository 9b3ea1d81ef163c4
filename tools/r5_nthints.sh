#!/bin/bash
# cache-policy hints one at a time (variant libraries built with -DSFG_NT_<X>, sfgwas_amd/lib/libsfgwas_hip_<X>.so) against the default build; 50k x 500k
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05nthints; mkdir -p $O
run() { local name=$1; shift
  env "$@" python3 bench.py --gpus 1 --config ${CFG:-c3} --steps 3 --warmup 2 --no-cpu-baseline --no-check > $O/$name.log 2>&1
  python3 - "$name" <<'PY'
import json, sys
d = json.loads([l for l in open(f"gpurun_out/r05nthints/{sys.argv[1]}.log") if l.startswith("{")][-1])
ph = d.get("phases_ms_per_step", {})
print(sys.argv[1], round(d["ms_per_step"]), d.get("digests", {}).get("out1_sha256", "")[:8], {k: round(v, 1) for k, v in ph.items() if k in ("skew", "encode", "mac_i8_pack_pt", "mac_small", "mac_big", "mac_i8_untile", "rotate")})
PY
}
run base
for v in ${VARIANTS:-FFT_D SKEW_ST PACK_LD PACK_ST MAC_DMA UNTILE}; do run $v SFG_LIB_PATH=$GRAFT_REPO_ROOT/sfgwas_amd/lib/libsfgwas_hip_$v.so; done
run base_b
