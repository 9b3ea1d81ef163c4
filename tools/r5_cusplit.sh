#!/bin/bash
# Round 5, VERDICT item 5: does partitioning the CUs between the fp64-issue-bound encode kernels and the HBM-bound transposition / MAC kernels pay?
# Part 1: every kernel's time against the number of CUs its queue may use (one queue, SFG_CU_MAIN = bits of hipExtStreamCreateWithCUMask).
# Part 2: the encode of MAC launch k + 1 on its own queue (SFG_MM_ENC_OVERLAP=1) restricted to E CUs, everything else on the other 256 - E.
# usage: tools/r5_cusplit.sh [config] ; writes gpurun_out/r05cu/*.json and table.txt
CFG=${1:-c3}
OUT=gpurun_out/r05cu; mkdir -p $OUT
run() {   # name, env...
  local name=$1; shift
  env "$@" SFG_BENCH_OWN_STREAM=1 python bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-check > $OUT/$name.json 2> $OUT/$name.err || echo "FAILED $name" >> $OUT/table.txt
}
: > $OUT/table.txt
run all
for n in 224 192 160 128 96 64; do run main_0-$n SFG_CU_MAIN=0-$n; done
run main_stride2 SFG_CU_MAIN=0-8,16-24,32-40,48-56,64-72,80-88,96-104,112-120,128-136,144-152,160-168,176-184,192-200,208-216,224-232,240-248
run ovl_nomask SFG_MM_ENC_OVERLAP=1
for e in 64 96 128 160; do run ovl_enc$e SFG_MM_ENC_OVERLAP=1 SFG_CU_ENC=0-$e SFG_CU_MAIN=$e-256; done
for e in 96 128; do run ovl_enc${e}_mainall SFG_MM_ENC_OVERLAP=1 SFG_CU_ENC=0-$e; done
python - <<'PY' >> gpurun_out/r05cu/table.txt
import json, glob, os
rows = []
for f in sorted(glob.glob("gpurun_out/r05cu/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception as e:
        rows.append((os.path.basename(f), "no line")); continue
    ph = d.get("phases_ms_per_step", {})
    rows.append((os.path.basename(f)[:-5], d["ms_per_step"], d["digests"]["out1_sha256"][:8], {k: round(v) for k, v in ph.items() if k in ("encode", "ntt_plain", "mac_small", "mac_big", "mac_i8_pack_pt", "mac_i8_untile", "rotate", "skew")}))
for r in rows:
    print(*r)
PY
cat $OUT/table.txt
