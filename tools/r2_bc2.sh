# quick parity subset + c3/c4 timing of the broadcast MAC
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${1:-bc2}; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_mac.py tests/test_gpu_matmul.py tests/test_gpu_properties.py -x -q -m gpu > $O/tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -5 $O/tests.log
[ $rc -ne 0 ] && exit $rc
for cfg in c3 c4; do for v in "SFG_X=0" "SFG_MM_NO_OVERLAP=1"; do
env $v timeout -k 10 600 python bench.py --config $cfg --no-cpu-baseline --no-check 2>&1 | grep "^{" > $O/bench_${cfg}_$v.json
python -c "
import json; r=json.load(open('$O/bench_${cfg}_$v.json')); p=r['phases_ms_per_step']; print('$cfg %-20s total %.0f  encode %.0f  mac_small %.0f  mac_big %.0f  rotate %.0f skew %.0f' % ('$v', r['ms_per_step'], p['encode'], p['mac_small'], p['mac_big'], p['rotate'], p['skew'])); print(r['digests']['out1_sha256'][:16], r['digests']['out2_sha256'][:16])"
done; done
