# the driver's command, as it will run it at round end
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04drv
( while sleep 50; do echo "tick $(date +%T)"; done ) & TICK=$!
timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04drv/bench.json 2> gpurun_out/r04drv/bench.err; rc=$?
kill $TICK
tail -3 gpurun_out/r04drv/bench.err
python - <<P
import json
r=json.load(open("gpurun_out/r04drv/bench.json"))
print(round(r["ms_per_step"]), r["value"], {k:round(x) for k,x in r["phases_ms_per_step"].items()})
print(r["digests"]["out1_sha256"][:8], r["digests"]["out2_sha256"][:8], r["parity_gate"], r["config"]["plaintext_cache"])
print({k:(v if not isinstance(v,dict) else '...') for k,v in r["roofline"].items() if k!="what"})
print(r["cpu_baseline"]["value"], r["cpu_baseline"]["cores"], r["cpu_baseline"]["cores_granted_vs_present"], r["cpu_baseline"]["sample"][-200:])
print(r["encoder_near_ties"])
P
exit $rc
