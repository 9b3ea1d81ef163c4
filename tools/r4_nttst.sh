cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/nttst
for st in 0 1 2 0 1 2; do
  SFG_LIB_PATH=$GRAFT_REPO_ROOT/sfgwas_amd/lib_ab/lib_st$st.so SFG_MM_NO_OVERLAP=1 timeout -k 10 300 python bench.py --config c2 --steps 3 --warmup 1 --no-cpu-baseline --no-check > gpurun_out/nttst/s$st.json 2> gpurun_out/nttst/s$st.err || { tail -5 gpurun_out/nttst/s$st.err; exit 1; }
  python - <<P
import json
r=json.load(open("gpurun_out/nttst/s$st.json"))
print("stagger $st", round(r["ms_per_step"]), r["roofline"].get("avg_launch_ms"), r["roofline"].get("kernel","")[:30], r["digests"]["out1_sha256"][:8])
P
done
