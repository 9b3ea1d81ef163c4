cd $GRAFT_REPO_ROOT; TAG=${1:-r03_stream}; mkdir -p gpurun_out/$TAG
timeout -k 10 1000 python3 tools/bench_stream.py --snps 131072 --dir $GRAFT_REPO_ROOT > gpurun_out/$TAG/bench_stream16.log 2>&1; rc=$?
tail -2 gpurun_out/$TAG/bench_stream16.log | cut -c1-2500
exit $rc
