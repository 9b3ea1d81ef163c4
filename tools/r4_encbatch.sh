cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/encbatch
for rep in 1 2; do for b in 1024 768 1536 2048; do
  SFG_ENC_BATCH=$b timeout -k 10 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-check --no-digest > gpurun_out/encbatch/b$b.json 2> gpurun_out/encbatch/b.err || { tail -5 gpurun_out/encbatch/b.err; exit 1; }
  python - <<P
import json
r=json.load(open("gpurun_out/encbatch/b$b.json"))
print("enc_batch $b", round(r["ms_per_step"]), round(r["phases_ms_per_step"]["encode"]), r["roofline"]["avg_launch_ms"])
P
done; done
