# quick: parity subset + c3 (single queue) + c4 timing
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${1:-q1}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_mac.py tests/test_gpu_matmul.py tests/test_gpu_encode.py tests/test_gpu_ntt.py tests/test_gpu_rotate.py tests/test_gpu_evalops.py tests/test_gpu_properties.py -x -q -m gpu > $O/tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -5 $O/tests.log
[ $rc -ne 0 ] && exit $rc
for run in "c3 SFG_MM_NO_OVERLAP=1" "c4 SFG_X=0"; do set -- $run
env $2 timeout -k 10 600 python bench.py --config $1 --no-cpu-baseline --no-check 2>&1 | grep "^{" > $O/bench_$1.json
python -c "
import json; r=json.load(open('$O/bench_$1.json')); p=r['phases_ms_per_step']; print('$1 %-20s total %.0f  encode %.0f  mac_small %.0f  mac_big %.0f  rotate %.0f skew %.0f' % ('$2', r['ms_per_step'], p['encode'], p['mac_small'], p['mac_big'], p['rotate'], p['skew'])); print(r['digests']['out1_sha256'][:16], r['digests']['out2_sha256'][:16])"
done
