# round 2: validate the DPP-broadcast MAC (parity tests), then same-box A/B against the 8x3-tile LDS-DMA kernel
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${1:-bc1}; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_mac.py tests/test_gpu_matmul.py tests/test_gpu_properties.py tests/test_gpu_fullsize.py tests/test_gpu_diagcache.py -x -q -m gpu > $O/tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -5 $O/tests.log
[ $rc -ne 0 ] && exit $rc
for v in "SFG_X=0" "SFG_MAC_IMPL=dma"; do
env $v timeout -k 10 600 python bench.py --config ${2:-c4} --no-cpu-baseline --no-check 2>&1 | grep "^{" > $O/bench_$v.json
python -c "
import json; r=json.load(open('$O/bench_$v.json')); p=r['phases_ms_per_step']; print('%-20s total %.0f  encode %.0f  mac_small %.0f  mac_big %.0f  rotate %.0f skew %.0f' % ('$v', r['ms_per_step'], p['encode'], p['mac_small'], p['mac_big'], p['rotate'], p['skew'])); print(r['digests'])"
done
