# round 2: DPP microbenchmark + default bench under rocprofv3 --kernel-trace --stats; summaries land in gpurun_out/r02
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r02a}
mkdir -p $R/gpurun_out/$TAG
cd $R
hipcc -O3 --offload-arch=gfx950 -o tools/ubench_dpp tools/ubench_dpp.hip 2>/dev/null
if [ -x tools/ubench_dpp ]; then timeout -k 10 120 tools/ubench_dpp > gpurun_out/$TAG/ubench_dpp.txt 2>&1; cat gpurun_out/$TAG/ubench_dpp.txt; fi
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/stats -o stats -- python3 bench.py > gpurun_out/$TAG/bench_stats.log 2>&1; rc=$?
grep '^{"metric"' gpurun_out/$TAG/bench_stats.log > gpurun_out/$TAG/benchline.json
find gpurun_out/$TAG -name "*kernel_trace.csv" -delete; find gpurun_out/$TAG -name "*.db" -delete
tail -3 gpurun_out/$TAG/bench_stats.log | cut -c1-1500
exit $rc
