# round 3: one c4 bench pass (no cpu baseline, no oracle gate), prints ms per step, digests, phases
cd $GRAFT_REPO_ROOT; TAG=${1:-r03_quick}; mkdir -p gpurun_out/$TAG
SFG_BENCH_PT_CACHE_GB=${PTC:-0} timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-check --steps 2 --warmup 1 $EXTRA > gpurun_out/$TAG/bench.log 2>&1 || { tail -5 gpurun_out/$TAG/bench.log; exit 1; }
grep '^{' gpurun_out/$TAG/bench.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['digests']['out1_sha256'][:12], d['digests']['out2_sha256'][:12], {k: round(v) for k, v in d['phases_ms_per_step'].items()}, d['roofline']['second_kernel'].get('launches'))"
