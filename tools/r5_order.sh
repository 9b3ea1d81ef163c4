#!/bin/bash
# A rank's step through the library engine (SFG_MGPU_SOLO) against the torch-issued sequence (SFG_BENCH_SOLO), and what the number of HIP streams alive in the process does to it
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05order; mkdir -p $O
export SFG_BENCH_PT_CACHE_GB=0
run() { local name=$1; shift
  env "$@" python3 bench.py --gpus 1 --config ${CFG:-c4} --steps 3 --warmup 3 --no-cpu-baseline --no-check --no-digest > $O/$name.log 2>&1
  python3 - "$name" <<'PY'
import json, sys
d = json.loads([l for l in open(f"gpurun_out/r05order/{sys.argv[1]}.log") if l.startswith("{")][-1])
ph = d.get("kernel_phases_ms_per_step") or d.get("phases_ms_per_step")
print(sys.argv[1], round(d["ms_per_step"]), {k: round(v) for k, v in ph.items() if k in ("encode", "skew", "mac", "rotate", "mac_small", "mac_i8_pack_pt")})
PY
}
run lib_0of8 SFG_MGPU_SOLO=0/8
run lib_0of8_ownq SFG_MGPU_SOLO=0/8 SFG_BENCH_OWN_STREAM=1
run torch_0of8 SFG_BENCH_SOLO=0/8
run lib_7of8 SFG_MGPU_SOLO=7/8
run lib_0of4 SFG_MGPU_SOLO=0/4
run torch_0of4 SFG_BENCH_SOLO=0/4
run lib_0of2 SFG_MGPU_SOLO=0/2
run torch_0of2 SFG_BENCH_SOLO=0/2
unset SFG_BENCH_PT_CACHE_GB
run lib_0of8_ptcache SFG_MGPU_SOLO=0/8
run lib_0of8_extra_stream SFG_MGPU_SOLO=0/8 SFG_MM_OVERLAP=1 SFG_MM_ENC_OVERLAP=1
