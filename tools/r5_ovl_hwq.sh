#!/bin/bash
# Two- / three-queue overlap of a product against the one-queue default, under different numbers of hardware queues (GPU_MAX_HW_QUEUES) and with the product on the
# context's own queue or on a torch stream: were the earlier "overlap does not pay" results the schedule's fault or the hardware-queue effect of profiles/r05_mgpu_queue_count.txt?
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05ovlhwq; mkdir -p $O
run() { local name=$1; shift
  env "$@" python3 bench.py --gpus 1 --config ${CFG:-c3} --steps 3 --warmup 2 --no-cpu-baseline --no-check > $O/$name.log 2>&1
  python3 - "$name" <<'PY'
import json, sys
d = json.loads([l for l in open(f"gpurun_out/r05ovlhwq/{sys.argv[1]}.log") if l.startswith("{")][-1])
ph = d["config"].get("kernel_phases_ms_per_step") or d.get("kernel_phases_ms_per_step") or {}
print(sys.argv[1], round(d["ms_per_step"]), d.get("digests", {}).get("out1_sha256", "")[:8], {k: round(v) for k, v in ph.items() if k in ("encode", "skew", "mac_small", "rotate", "mac_i8_pack_pt")})
PY
}
OV="SFG_MM_OVERLAP=1 SFG_MM_ENC_OVERLAP=1"
run one
run two $OV
run mm SFG_MM_OVERLAP=1
for q in 2 3; do
  run one_hwq$q GPU_MAX_HW_QUEUES=$q
  run two_hwq$q $OV GPU_MAX_HW_QUEUES=$q
  run mm_hwq$q SFG_MM_OVERLAP=1 GPU_MAX_HW_QUEUES=$q
done
run two_ownq $OV SFG_BENCH_OWN_STREAM=1
run two_ownq_hwq2 $OV SFG_BENCH_OWN_STREAM=1 GPU_MAX_HW_QUEUES=2
run two_ownq_hwq3 $OV SFG_BENCH_OWN_STREAM=1 GPU_MAX_HW_QUEUES=3
run two_hwq8 $OV GPU_MAX_HW_QUEUES=8
