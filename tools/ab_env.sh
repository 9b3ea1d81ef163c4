# same-box A/B of runtime switches: usage  bash tools/ab_env.sh <config> "VAR=a" "VAR=b" ...   (each variant run twice, interleaved)
cd ${GRAFT_REPO_ROOT:-.}
cfg=$1; shift
for rep in 1 2; do for v in "$@"; do
env $v SFG_MM_NO_OVERLAP=1 timeout 600 python bench.py --config $cfg --no-cpu-baseline 2>&1 | grep "^{" > /tmp/o.json
python -c "
import json; r=json.load(open('/tmp/o.json')); p=r['phases_ms_per_step']; print('%-28s total %.0f  encode %.0f  mac_small %.0f  mac_big %.0f  rotate %.0f' % ('$v', r['ms_per_step'], p['encode'], p['mac_small'], p['mac_big'], p['rotate']))"
done; done
