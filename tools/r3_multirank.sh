# round 3: the N > 1 path on one GPU (gloo rehearsal) - tests/test_gpu_multirank.py
cd $GRAFT_REPO_ROOT; TAG=${1:-r03_mr}; mkdir -p gpurun_out/$TAG
timeout -k 10 1000 python -m pytest tests/test_gpu_multirank.py -x -q -m gpu --durations=0 > gpurun_out/$TAG/pytest.log 2>&1; rc=$?
tail -30 gpurun_out/$TAG/pytest.log
exit $rc
