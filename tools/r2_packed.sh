cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${1:-packed}; mkdir -p $O
for v in "" "--packed-geno"; do
timeout -k 10 600 python bench.py --config c4 --no-cpu-baseline --no-check $v 2>&1 | grep "^{" > $O/b.json
python -c "
import json; r=json.load(open('$O/b.json')); p=r['phases_ms_per_step']; print('%-14s total %.0f  encode %.0f  mac %.0f  skew %.0f  %s %s' % ('$v' or 'int8', r['ms_per_step'], p['encode'], p['mac'], p['skew'], r['digests']['out1_sha256'][:12], r['digests']['out2_sha256'][:12]))" | tee -a $O/packed.txt
done
