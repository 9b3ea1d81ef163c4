#!/bin/bash
# gpurun -- bash tools/r6_pmc_ride.sh [tag] [config] : fabric-side read and write bytes (separate rocprofv3 --pmc passes, as tools/pmc_ntt.sh) of the kernels of a step with the
# riding transposition - k_ntt_half3_move above all - at the config whose MAC-group shape the bench line is quoted on (the movers' bytes per launch depend on it: c4).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r06pmc}; CFG=${2:-c4}; mkdir -p $R/gpurun_out/$TAG; cd $R
make -C oracle native > /dev/null 2>&1
( while sleep 45; do echo "tick $(date +%T)"; done ) & TICK=$!
export SFG_UPLOAD_BLOCKING=1
rc=0
for p in "R TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum" "W TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  set -- $p; pn=$1; shift
  d=gpurun_out/$TAG/$pn; rm -rf $d
  timeout -k 10 ${TMO:-500} rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $d -o p -- python3 bench.py --config $CFG --steps 1 --warmup 0 --no-cpu-baseline --no-check --no-digest > gpurun_out/$TAG/$pn.log 2>&1 || { rc=$?; tail -5 gpurun_out/$TAG/$pn.log; break; }
  echo "$pn done"
done
kill $TICK
python3 tools/pmc_ntt.py gpurun_out/$TAG $CFG > gpurun_out/$TAG/summary.txt 2>&1
find gpurun_out/$TAG -name "*.csv" -size +20M -delete
cat gpurun_out/$TAG/summary.txt
exit $rc
