cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04stage
for v in 1 2; do
  SFG_MAC_I8_STAGE=$v timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check > gpurun_out/r04stage/bench_s$v.json 2> gpurun_out/r04stage/bench_s$v.err || { tail -5 gpurun_out/r04stage/bench_s$v.err; exit 1; }
  python - <<P
import json
r=json.load(open("gpurun_out/r04stage/bench_s$v.json"))
print("stage=$v", round(r["ms_per_step"]), {k:round(x) for k,x in r["phases_ms_per_step"].items()}, r["digests"]["out1_sha256"][:8], r["digests"]["out2_sha256"][:8])
P
done
