# first GPU call of round 4: the c4 test alone (timing printed), then the int8-MAC counter passes
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k c4_100000 -s > gpurun_out/r04_c4_test.log 2>&1; rc=$?
tail -15 gpurun_out/r04_c4_test.log
[ $rc = 0 ] || exit $rc
bash tools/r4_pmc_mac_i8.sh r04pmc c2
