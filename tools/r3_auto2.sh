cd $GRAFT_REPO_ROOT; TAG=${1:-r03_auto2}; mkdir -p gpurun_out/$TAG
for mode in packed_auto int8_auto; do
  flag=""; [ $mode = packed_auto ] && flag="--packed-geno"
  timeout -k 10 500 python3 bench.py $flag --steps 2 --warmup 1 --no-cpu-baseline --no-check > gpurun_out/$TAG/$mode.log 2>&1 || { tail -8 gpurun_out/$TAG/$mode.log; exit 1; }
  python3 - <<PY
import json
d=json.loads([l for l in open("gpurun_out/$TAG/$mode.log") if l.startswith("{")][-1])
print("$mode", "ms_per_step", round(d["ms_per_step"],1), "digests", d["digests"]["out1_sha256"][:12], d["digests"]["out2_sha256"][:12], {k: round(v) for k,v in d["phases_ms_per_step"].items()}, "mac launches", d["roofline"]["launches"])
PY
done | tee gpurun_out/$TAG/summary.txt
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_properties.py tests/test_gpu_matmul.py tests/test_gpu_packed.py -x -q -m gpu > gpurun_out/$TAG/pytest.log 2>&1; rc=$?; tail -3 gpurun_out/$TAG/pytest.log; exit $rc
