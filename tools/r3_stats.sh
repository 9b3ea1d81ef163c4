# round 3: the default bench under rocprofv3 --kernel-trace --stats (no counter passes); summaries land in gpurun_out/<tag>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r03_final}
mkdir -p $R/gpurun_out/$TAG
cd $R
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/stats -o stats -- python3 bench.py > gpurun_out/$TAG/bench_stats.log 2>&1 || { tail -5 gpurun_out/$TAG/bench_stats.log; exit 1; }
grep '^{"metric"' gpurun_out/$TAG/bench_stats.log > gpurun_out/$TAG/benchline.json
cp $(find gpurun_out/$TAG/stats -name '*kernel_stats.csv' | head -1) gpurun_out/$TAG/kernel_stats.csv
rm -rf gpurun_out/$TAG/stats
cut -c1-600 gpurun_out/$TAG/benchline.json; head -8 gpurun_out/$TAG/kernel_stats.csv | cut -c1-60,150-300
