# timing diagnostics of k_mac_i8_ring at c3 (50k x 500k: same launch shapes, 1/4 of the work): default / matrix pipe nearly idle / memory system idle
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04diag
timeout -k 10 300 python -m pytest tests/test_gpu_encode.py -x -q -m gpu > gpurun_out/r04diag/enc_tests.log 2>&1; tail -3 gpurun_out/r04diag/enc_tests.log
for v in 0 1 2; do
  SFG_MAC_I8_DIAG=$v SFG_MM_NO_OVERLAP=1 timeout -k 10 300 python bench.py --config c3 --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-digest > gpurun_out/r04diag/b$v.json 2> gpurun_out/r04diag/b$v.err || { tail -5 gpurun_out/r04diag/b$v.err; exit 1; }
  python - <<P
import json
r=json.load(open("gpurun_out/r04diag/b$v.json"))
print("diag=$v", round(r["ms_per_step"]), {k:round(x) for k,x in r["phases_ms_per_step"].items() if k.startswith("mac")}, "launch ms", r["roofline"]["second_kernel"].get("avg_launch_ms") if "second_kernel" in r["roofline"] else r["roofline"].get("avg_launch_ms"))
P
done
