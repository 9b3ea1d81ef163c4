#!/bin/bash
# gpurun -- bash tools/r6_compact.sh : compact panel rows.  Product tests vs the oracle, then c3 / c4 steps in the A/B build with SFG_PT_COMPACT=0 / 1.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06compact; mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 700 python3 -m pytest tests/test_gpu_matmul.py tests/test_gpu_properties.py tests/test_gpu_edge.py tests/test_gpu_ptcache.py tests/test_gpu_fullsize.py tests/test_gpu_mac_i8.py tests/test_gpu_mgpu.py -m gpu -x -q -k "not c4_100000" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
fi
AB=$GRAFT_REPO_ROOT/sfgwas_amd/lib_ab/libsfgwas_hip.so
run() { local name=$1; shift
  env SFG_LIB_PATH=$AB "$@" timeout -k 10 500 python3 bench.py --gpus 1 --config ${CFG:-c3} --steps ${STEPS:-3} --warmup 2 --no-cpu-baseline --no-check > $O/$name.log 2>&1 || { tail -5 $O/$name.log; return 1; }
  python3 - "$O/$name.log" "$name" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
ph = d.get("phases_ms_per_step", {})
r = d.get("roofline", {})
print(sys.argv[2], round(d["ms_per_step"]), d.get("digests_match_pinned"), {k: round(v, 1) for k, v in ph.items() if k in ("encode", "mac_i8_pack_pt", "mac_small", "mac_big", "mac_i8_untile")}, r.get("kernel"), round(r.get("avg_launch_ms", 0), 4), r.get("launches"), "mac launches", r.get("second_kernel", {}).get("launches"))
PY
}
for cfg in ${CFGS:-c3 c4}; do for v in ${VARS:-"SFG_PT_COMPACT=0" "SFG_PT_COMPACT=1"}; do CFG=$cfg run ${cfg}_$v $v || exit 1; done; done
