# round 3: run a set of GPU test files; log under gpurun_out/<tag>/
cd $GRAFT_REPO_ROOT; TAG=$1; shift; mkdir -p gpurun_out/$TAG
timeout -k 10 ${TMO:-1000} python -m pytest "$@" -x -q -m gpu --durations=15 > gpurun_out/$TAG/pytest.log 2>&1; rc=$?
tail -40 gpurun_out/$TAG/pytest.log
exit $rc
