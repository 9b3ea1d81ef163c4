cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r01f
cd $R
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01f/stats -o stats -- python3 bench.py > gpurun_out/r01f/bench_stats.log 2>&1
grep '^{"metric"' gpurun_out/r01f/bench_stats.log > gpurun_out/r01f/benchline.json



# keep only the small summaries
find gpurun_out/r01f -name "*kernel_trace.csv" -delete; find gpurun_out/r01f -name "*counter_collection.csv" -delete; find gpurun_out/r01f -name "*.db" -delete
find gpurun_out/r01f -type f | head -40
tail -3 gpurun_out/r01f/traffic.txt
