# generic same-box A/B of two library builds at c4: bash tools/r4_ab_env.sh <tag> <libA> <libB> [tests...]
cd $GRAFT_REPO_ROOT; TAG=$1; A=$2; B=$3; shift 3; mkdir -p gpurun_out/$TAG
if [ $# -gt 0 ]; then timeout -k 10 600 python -m pytest "$@" -x -q -m gpu > gpurun_out/$TAG/tests.log 2>&1; rc=$?; tail -3 gpurun_out/$TAG/tests.log; [ $rc = 0 ] || exit $rc; fi
for lib in $A $B $A $B; do
  name=$(basename $lib .so)
  SFG_LIB_PATH=$GRAFT_REPO_ROOT/$lib SFG_MM_NO_OVERLAP=1 timeout -k 10 400 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check > gpurun_out/$TAG/$name.json 2> gpurun_out/$TAG/$name.err || { tail -5 gpurun_out/$TAG/$name.err; exit 1; }
  python - <<P
import json
r=json.load(open("gpurun_out/$TAG/$name.json"))
print("$name", round(r["ms_per_step"]), {k:round(x) for k,x in r["phases_ms_per_step"].items()}, r["digests"]["out1_sha256"][:8], r["digests"]["out2_sha256"][:8])
P
done
