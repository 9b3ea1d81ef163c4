# same-box A/B of two library builds: bash tools/r2_ab_lib.sh <tag> <config> <libA> <libB> ...   (each twice, interleaved; single queue)
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/$1; cfg=$2; shift 2; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_ntt.py tests/test_gpu_encode.py tests/test_gpu_rotate.py tests/test_gpu_evalops.py tests/test_gpu_matmul.py tests/test_gpu_mac.py -x -q -m gpu > $O/tests.log 2>&1; rc=$?
echo "tests rc=$rc"; tail -3 $O/tests.log
[ $rc -ne 0 ] && exit $rc
for rep in 1 2; do for lib in "$@"; do
SFG_LIB_PATH=$PWD/$lib SFG_MM_NO_OVERLAP=1 timeout -k 10 600 python bench.py --config $cfg --no-cpu-baseline --no-check 2>&1 | grep "^{" > $O/b.json
python -c "
import json; r=json.load(open('$O/b.json')); p=r['phases_ms_per_step']; print('%-44s total %.0f  encode %.0f  mac_small %.0f  mac_big %.0f  rotate %.0f  %s' % ('$lib', r['ms_per_step'], p['encode'], p['mac_small'], p['mac_big'], p['rotate'], r['digests']['out1_sha256'][:12]))" | tee -a $O/ab.txt
done; done
