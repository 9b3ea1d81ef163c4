# bench lines of the smaller BASELINE configs (parity-test cases; recorded for reference), plus the collectives path at world size 1
cd ${GRAFT_REPO_ROOT:-.}
O=gpurun_out/${1:-cfgs}; mkdir -p $O
for cfg in c2 c3; do
timeout -k 10 600 python bench.py --config $cfg --steps 3 --warmup 1 2>&1 | grep "^{" > $O/bench_$cfg.json
python -c "
import json; r=json.load(open('$O/bench_$cfg.json')); print('$cfg', round(r['ms_per_step'],1), 'ms/step', '%.3e' % r['value'], r['parity_gate']['status'], r['phases_ms_per_step'])"
done
SFG_BENCH_FORCE_COLLECTIVES=1 timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --config c3 --no-cpu-baseline 2>&1 | grep "^{" > $O/bench_c3_collectives_w1.json
python -c "
import json; a=json.load(open('$O/bench_c3.json')); b=json.load(open('$O/bench_c3_collectives_w1.json')); print('c3 collectives path', round(b['ms_per_step'],1), 'ms/step; digests equal:', a['digests']['out1_sha256']==b['digests']['out1_sha256'] and a['digests']['out2_sha256']==b['digests']['out2_sha256'])"
