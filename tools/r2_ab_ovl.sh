# same-box A/B of library builds WITH the two-queue overlap (the default schedule): bash tools/r2_ab_ovl.sh <tag> <config> <lib> ...
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/$1; cfg=$2; shift 2; mkdir -p $O
for rep in 1 2; do for lib in "$@"; do
SFG_LIB_PATH=$PWD/$lib timeout -k 10 600 python bench.py --config $cfg --no-cpu-baseline --no-check 2>&1 | grep "^{" > $O/b.json
python -c "
import json; r=json.load(open('$O/b.json')); p=r['phases_ms_per_step']; print('%-44s total %.0f  encode %.0f  mac_small %.0f  mac_big %.0f  rotate %.0f  %s' % ('$lib', r['ms_per_step'], p['encode'], p['mac_small'], p['mac_big'], p['rotate'], r['digests']['out1_sha256'][:12]))" | tee -a $O/ab.txt
done; done
