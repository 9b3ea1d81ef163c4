# round 3: streamed association path - tests + the config-5 shaped batch measurement (rotation cache per call vs per batch, O_DIRECT reads)
cd $GRAFT_REPO_ROOT; TAG=${1:-r03_stream}; mkdir -p gpurun_out/$TAG
timeout -k 10 600 python -m pytest tests/test_gpu_stream.py -x -q -m gpu > gpurun_out/$TAG/pytest.log 2>&1 || { tail -30 gpurun_out/$TAG/pytest.log; exit 1; }
tail -3 gpurun_out/$TAG/pytest.log
df -h . /tmp | tail -3
timeout -k 10 900 python3 tools/bench_stream.py --dir $GRAFT_REPO_ROOT > gpurun_out/$TAG/bench_stream.log 2>&1; rc=$?
tail -2 gpurun_out/$TAG/bench_stream.log | cut -c1-2500
exit $rc
