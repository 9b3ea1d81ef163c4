# round 3: default bench under rocprofv3 --kernel-trace --stats, then the two PMC traffic passes; summaries land in gpurun_out/<tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r03}
mkdir -p $R/gpurun_out/$TAG
cd $R
timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/stats -o stats -- python3 bench.py > gpurun_out/$TAG/bench_stats.log 2>&1 || { tail -5 gpurun_out/$TAG/bench_stats.log; exit 1; }
grep '^{"metric"' gpurun_out/$TAG/bench_stats.log > gpurun_out/$TAG/benchline.json
find gpurun_out/$TAG -name "*kernel_trace.csv" -delete; find gpurun_out/$TAG -name "*.db" -delete
cut -c1-700 gpurun_out/$TAG/benchline.json
export SFG_MM_NO_OVERLAP=1 SFG_UPLOAD_BLOCKING=1      # the counter passes serialise dispatches: single queue, blocking uploads (traffic per MAC launch is unaffected)
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/$TAG/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-check --no-digest > gpurun_out/$TAG/pmc_fetch.log 2>&1 || { tail -5 gpurun_out/$TAG/pmc_fetch.log; exit 1; }
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/$TAG/pmc_write -o w -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-check --no-digest > gpurun_out/$TAG/pmc_write.log 2>&1 || { tail -5 gpurun_out/$TAG/pmc_write.log; exit 1; }
python3 tools/pmc_traffic.py gpurun_out/$TAG/pmc_fetch gpurun_out/$TAG/pmc_write gpurun_out/$TAG/traffic.json > gpurun_out/$TAG/traffic.txt 2>&1
find gpurun_out/$TAG -name "*counter_collection.csv" -delete
head -4 gpurun_out/$TAG/traffic.txt
