# diagnostic build only (-DSFG_MAC_DIAG): where does the broadcast MAC's time go?  c3, single queue
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${1:-diag1}; mkdir -p $O
for dg in 0 1 2 3 4 7 8 12; do
SFG_MAC_DIAG=$dg SFG_MM_NO_OVERLAP=1 timeout -k 10 300 python bench.py --config c3 --no-cpu-baseline --no-check --no-digest 2>&1 | grep "^{" > $O/b.json
python -c "
import json; r=json.load(open('$O/b.json')); p=r['phases_ms_per_step']; print('diag %-3s total %.0f  encode %.0f  mac_small %.0f  mac_big %.0f  rotate %.0f' % ('$dg', r['ms_per_step'], p['encode'], p['mac_small'], p['mac_big'], p['rotate']))" | tee -a $O/diag.txt
done
