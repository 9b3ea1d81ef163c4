cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/nttpad
for pad in 0 6000 18432 45000; do
  SFG_NTT_LDS_PAD=$pad SFG_LIB_PATH=$GRAFT_REPO_ROOT/sfgwas_amd/lib_ab/lib_nttpad.so SFG_MM_NO_OVERLAP=1 timeout -k 10 300 python bench.py --config c2 --steps 3 --warmup 1 --no-cpu-baseline --no-check > gpurun_out/nttpad/p$pad.json 2> gpurun_out/nttpad/p$pad.err || { tail -5 gpurun_out/nttpad/p$pad.err; exit 1; }
  python - <<P
import json
r=json.load(open("gpurun_out/nttpad/p$pad.json"))
print("pad $pad", round(r["ms_per_step"]), {k:round(x) for k,x in r["phases_ms_per_step"].items() if k in ("encode","ntt_plain","mac")}, r["roofline"].get("avg_launch_ms"), r["roofline"].get("kernel","")[:30])
P
done
