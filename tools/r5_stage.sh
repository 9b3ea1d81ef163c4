#!/bin/bash
# Round 5, VERDICT r4 item 4: the streamed transposition (SFG_MAC_I8_STAGE=1: no plaintext panel, no transposition pass) with batches that complete whole 16-column tiles
# (SFG_STAGE_GIANTS=16) and on the product's own queue (SFG_STAGE_SAMEQ=1), against the default (panel + pass) and round 4's form (11 giants, encode queue)
CFG=${1:-c3}; OUT=gpurun_out/r05stage; mkdir -p $OUT; : > $OUT/table.txt
run() { local name=$1; shift
  env "$@" python bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-check > $OUT/$name.json 2> $OUT/$name.err || echo "FAILED $name" >> $OUT/table.txt; }
run default
run stage_g11_encq SFG_MAC_I8_STAGE=1
run stage_g16_encq SFG_MAC_I8_STAGE=1 SFG_STAGE_GIANTS=16
run stage_g16_sameq SFG_MAC_I8_STAGE=1 SFG_STAGE_GIANTS=16 SFG_STAGE_SAMEQ=1
run stage_g11_sameq SFG_MAC_I8_STAGE=1 SFG_STAGE_SAMEQ=1
run stage_g32_sameq SFG_MAC_I8_STAGE=1 SFG_STAGE_GIANTS=32 SFG_STAGE_SAMEQ=1
python - <<'PY' >> gpurun_out/r05stage/table.txt
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r05stage/*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
    except Exception:
        print(os.path.basename(f), "no line"); continue
    ph = d.get("phases_ms_per_step", {})
    print(os.path.basename(f)[:-5], round(d["ms_per_step"]), d["digests"]["out1_sha256"][:8], d["digests"]["out2_sha256"][:8], {k: round(v) for k, v in ph.items() if k in ("encode", "mac_small", "mac_big", "mac_i8_pack_pt", "mac_i8_untile", "rotate")})
PY
cat $OUT/table.txt
