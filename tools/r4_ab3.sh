# same-box A/B of several library builds at c4: bash tools/r4_ab3.sh <tag> <lib>...
cd $GRAFT_REPO_ROOT; TAG=$1; shift; mkdir -p gpurun_out/$TAG
for lib in "$@" "$@"; do
  name=$(basename $lib .so)
  SFG_LIB_PATH=$GRAFT_REPO_ROOT/$lib SFG_MM_NO_OVERLAP=1 timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err || { tail -5 gpurun_out/$TAG/$name.err; exit 1; }
  python - <<P
import json
r=json.load(open("gpurun_out/$TAG/$name.json"))
print("$name", round(r["ms_per_step"]), {k:round(x) for k,x in r["phases_ms_per_step"].items()}, r["digests"]["out1_sha256"][:8], r["digests"]["out2_sha256"][:8])
P
done
