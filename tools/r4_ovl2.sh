cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04ovl2
for cfg in c4 c3; do for v in 0 1 0 1; do
  if [ $v = 1 ]; then unset SFG_MM_OVERLAP; else export SFG_MM_OVERLAP=1; fi      # v = 1: one queue (the default since round 4), v = 0: the two-queue overlap
  timeout -k 10 400 python bench.py --config $cfg --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-digest > gpurun_out/r04ovl2/b_${cfg}_$v.json 2> gpurun_out/r04ovl2/b.err || { tail -5 gpurun_out/r04ovl2/b.err; exit 1; }
  python - <<P
import json
r=json.load(open("gpurun_out/r04ovl2/b_${cfg}_$v.json"))
print("$cfg no_overlap=$v", round(r["ms_per_step"]))
P
done; done
