cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/p2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_mac.py tests/test_gpu_matmul.py tests/test_gpu_properties.py "tests/test_gpu_fullsize.py::test_full_block_all_8192_diagonals_vs_oracle" "tests/test_gpu_fullsize.py::test_multi_group_two_pass_overlap_vs_oracle" -x -q > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
for rep in 1 2; do for v in "SFG_X=0" "SFG_MAC_PT=plain"; do
env $v timeout 600 python bench.py --config c3 --no-cpu-baseline --no-check --no-digest 2>&1 | grep "^{" > /tmp/o.json
python -c "
import json; r=json.load(open('/tmp/o.json')); p=r['phases_ms_per_step']; print('%-20s total %.0f  encode %.0f  mac_small %.0f  mac_big %.0f  rotate %.0f skew %.0f' % ('$v', r['ms_per_step'], p['encode'], p['mac_small'], p['mac_big'], p['rotate'], p['skew']))"
done; done
