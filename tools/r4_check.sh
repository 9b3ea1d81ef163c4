# default build at c4 (quick): expect ~11.1 s per step as in gpurun_out/r04big (46-bit modulus on the ring MAC)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04chk
timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check > gpurun_out/r04chk/bench.json 2> gpurun_out/r04chk/bench.err || { tail -5 gpurun_out/r04chk/bench.err; exit 1; }
python - <<P
import json
r=json.load(open("gpurun_out/r04chk/bench.json"))
print("default", round(r["ms_per_step"]), {k:round(x) for k,x in r["phases_ms_per_step"].items()}, r["digests"]["out1_sha256"][:8], r["digests"]["out2_sha256"][:8], r["config"]["plaintext_cache"][:60])
P
