# round 3: the whole GPU test suite, one process, log under gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT; TAG=${1:-r03_suite}; mkdir -p gpurun_out/$TAG
timeout -k 10 1150 python -m pytest tests -x -q -m gpu --durations=25 -p no:cacheprovider > gpurun_out/$TAG/pytest.log 2>&1; rc=$?
tail -45 gpurun_out/$TAG/pytest.log
exit $rc
