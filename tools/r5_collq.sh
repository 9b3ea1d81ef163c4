#!/bin/bash
# What exactly slows the kernels of a rank's step when the engine has a collectives queue of its own (profiles/r05_mgpu_queue_count.txt, part 1)?
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05collq; mkdir -p $O
export SFG_BENCH_PT_CACHE_GB=0
run() { local name=$1; shift
  env "$@" python3 bench.py --gpus 1 --config ${CFG:-c4} --steps 3 --warmup 3 --no-cpu-baseline --no-check --no-digest > $O/$name.log 2>&1
  python3 - "$name" <<'PY'
import json, sys
d = json.loads([l for l in open(f"gpurun_out/r05collq/{sys.argv[1]}.log") if l.startswith("{")][-1])
ph = d.get("kernel_phases_ms_per_step") or d.get("phases_ms_per_step")
print(sys.argv[1], round(d["ms_per_step"]), {k: round(v) for k, v in ph.items() if k in ("encode", "skew", "mac", "rotate")})
PY
}
S=SFG_MGPU_SOLO=0/8
run enc_ownq $S SFG_BENCH_OWN_STREAM=1
run own_ownq $S SFG_BENCH_OWN_STREAM=1 SFG_MGPU_COLL_QUEUE=own
run spare_ownq $S SFG_BENCH_OWN_STREAM=1 SFG_MGPU_COLL_QUEUE=spare
run own_torchq $S SFG_MGPU_COLL_QUEUE=own
run own_ownq_hwq2 $S SFG_BENCH_OWN_STREAM=1 SFG_MGPU_COLL_QUEUE=own GPU_MAX_HW_QUEUES=2
run own_ownq_hwq3 $S SFG_BENCH_OWN_STREAM=1 SFG_MGPU_COLL_QUEUE=own GPU_MAX_HW_QUEUES=3
run own_ownq_hwq5 $S SFG_BENCH_OWN_STREAM=1 SFG_MGPU_COLL_QUEUE=own GPU_MAX_HW_QUEUES=5
CFG=c2 run c2_enc_ownq SFG_MGPU_SOLO=0/2 SFG_BENCH_OWN_STREAM=1
CFG=c2 run c2_own_ownq SFG_MGPU_SOLO=0/2 SFG_BENCH_OWN_STREAM=1 SFG_MGPU_COLL_QUEUE=own
