#!/bin/bash
# Part 2 of tools/r5_cusplit.sh with the switches that actually enable the third queue (SFG_MM_OVERLAP=1 is a precondition of SFG_MM_ENC_OVERLAP=1)
CFG=${1:-c3}
OUT=gpurun_out/r05cu2; mkdir -p $OUT
run() { local name=$1; shift
  env "$@" SFG_BENCH_OWN_STREAM=1 python bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-check > $OUT/$name.json 2> $OUT/$name.err || echo "FAILED $name" >> $OUT/table.txt; }
: > $OUT/table.txt
run one_queue
run two_queues SFG_MM_OVERLAP=1
run three_queues SFG_MM_OVERLAP=1 SFG_MM_ENC_OVERLAP=1
for e in 64 96 128; do
  run three_enc${e}_rest SFG_MM_OVERLAP=1 SFG_MM_ENC_OVERLAP=1 SFG_CU_ENC=0-$e SFG_CU_MAIN=$e-256 SFG_CU_AUX=$e-256
  run three_enc${e}_mainall SFG_MM_OVERLAP=1 SFG_MM_ENC_OVERLAP=1 SFG_CU_ENC=0-$e
done
run three_enc96_main160_auxenc SFG_MM_OVERLAP=1 SFG_MM_ENC_OVERLAP=1 SFG_CU_ENC=0-96 SFG_CU_MAIN=96-256 SFG_CU_AUX=0-96
python - <<'PY' >> gpurun_out/r05cu2/table.txt
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r05cu2/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        print(os.path.basename(f), "no line"); continue
    ph = d.get("phases_ms_per_step", {})
    print(os.path.basename(f)[:-5], round(d["ms_per_step"]), d["digests"]["out1_sha256"][:8], {k: round(v) for k, v in ph.items() if k in ("encode", "mac_small", "mac_big", "mac_i8_pack_pt", "mac_i8_untile", "rotate")})
PY
cat $OUT/table.txt
