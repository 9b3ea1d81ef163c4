#!/bin/bash
# Two-queue overlap (SFG_MM_OVERLAP=1 SFG_MM_ENC_OVERLAP=1) against the one-queue default for one rank of a 2 / 4 / 8-rank step through the library engine
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05ovl; mkdir -p $O
export SFG_BENCH_PT_CACHE_GB=0
run() { local name=$1; shift
  env "$@" python3 bench.py --gpus 1 --config ${CFG:-c4} --steps 3 --warmup 3 --no-cpu-baseline --no-check --no-digest > $O/$name.log 2>&1
  python3 - "$name" <<'PY'
import json, sys
d = json.loads([l for l in open(f"gpurun_out/r05ovl/{sys.argv[1]}.log") if l.startswith("{")][-1])
print(sys.argv[1], round(d["ms_per_step"]))
PY
}
OV="SFG_MM_OVERLAP=1 SFG_MM_ENC_OVERLAP=1"
run one_0of8 SFG_MGPU_SOLO=0/8
run two_0of8 SFG_MGPU_SOLO=0/8 $OV
run one_7of8 SFG_MGPU_SOLO=7/8
run two_7of8 SFG_MGPU_SOLO=7/8 $OV
run two_0of8_b SFG_MGPU_SOLO=0/8 $OV
run one_0of8_b SFG_MGPU_SOLO=0/8
run one_0of4 SFG_MGPU_SOLO=0/4
run two_0of4 SFG_MGPU_SOLO=0/4 $OV
run two_0of2 SFG_MGPU_SOLO=0/2 $OV
run two_0of8_mmonly SFG_MGPU_SOLO=0/8 SFG_MM_OVERLAP=1
