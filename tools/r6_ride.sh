#!/bin/bash
# gpurun -- bash tools/r6_ride.sh : the riding transposition (kernels.hpp PtRide).  (1) product tests against the oracle with it on (the product library's default),
# (2) 50k x 500k / 100k x 1M steps in the A/B build: SFG_PT_RIDE=0 (the pass before every MAC launch) against riding with 192 / 128 / 256 mover workgroups per NTT launch.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06ride; mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 700 python3 -m pytest tests/test_gpu_matmul.py tests/test_gpu_properties.py tests/test_gpu_edge.py tests/test_gpu_ptcache.py tests/test_gpu_fullsize.py -m gpu -x -q -k "not c4_100000" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
fi
AB=$GRAFT_REPO_ROOT/sfgwas_amd/lib_ab/libsfgwas_hip.so
run() { local name=$1; shift
  env SFG_LIB_PATH=$AB "$@" timeout -k 10 500 python3 bench.py --gpus 1 --config ${CFG:-c3} --steps ${STEPS:-3} --warmup 2 --no-cpu-baseline --no-check > $O/$name.log 2>&1 || { tail -5 $O/$name.log; return 1; }
  python3 - "$O/$name.log" "$name" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
ph = d.get("phases_ms_per_step", {})
print(sys.argv[2], round(d["ms_per_step"]), d.get("digests", {}).get("out1_sha256", "")[:8], d.get("digests", {}).get("out2_sha256", "")[:8], d.get("digests_match_pinned"), {k: round(v, 1) for k, v in ph.items() if k in ("encode", "mac_i8_pack_pt", "mac_small", "mac_big", "rotate")})
PY
}
for v in ${VARIANTS:-0 192 128 256 0}; do run ride_$v SFG_PT_RIDE=$v || exit 1; done
