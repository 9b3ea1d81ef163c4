cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/p3; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_diagcache.py tests/test_host_mirror.py tests/test_gpu_encode.py tests/test_gpu_evalops.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?"; tail -15 $O/tests.log
