# round 2: full GPU parity suite, log kept under gpurun_out/r02
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/r02; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=25 > $O/gpu_tests.log 2>&1; rc=$?
tail -40 $O/gpu_tests.log
exit $rc
