cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04suite
( while sleep 50; do echo "tick $(date +%T)"; done ) & TICK=$!
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu --durations=15 > gpurun_out/r04suite/tests.log 2>&1; rc=$?
kill $TICK
tail -25 gpurun_out/r04suite/tests.log
exit $rc
