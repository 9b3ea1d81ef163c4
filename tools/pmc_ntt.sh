#!/bin/bash
# Round 5: exact fabric-side traffic of the dominant kernel k_ntt_half3<false, true> at its product launch shape (2048 plaintexts x 5 moduli since the streaming stores; 1024 before:
# the same at every config; run at c2) - read AND write requests in separate counter passes - plus two SQ passes (issue / wait breakdown).
#   R  TCC_EA0_RDREQ_{32B,64B,128B}_sum TCC_EA0_RDREQ_sum      read bytes  = 32 n32 + 64 n64 + 128 n128   (exact: no FETCH_SIZE "x 2" correction)
#   W  TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum                 write bytes = 64 n64 + 32 (n - n64)
#   S1 / S2  SQ counters
# Counter passes only (no --kernel-trace domain mixes): bench.py runs with --no-cpu-baseline (no make / gcc children under the profiler's preload).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r05pmc}; CFG=${2:-c2}; mkdir -p $R/gpurun_out/$TAG; cd $R
make -C oracle native > /dev/null 2>&1
( while sleep 45; do echo "tick $(date +%T)"; done ) & TICK=$!
export SFG_UPLOAD_BLOCKING=1
rc=0
for p in "R TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum" "W TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
         "S1 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES GRBM_GUI_ACTIVE" \
         "S2 SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM"; do
  set -- $p; pn=$1; shift
  d=gpurun_out/$TAG/$pn; rm -rf $d
  timeout -k 10 ${TMO:-400} rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $d -o p -- python3 bench.py --config $CFG --steps 1 --warmup 0 --no-cpu-baseline --no-check --no-digest > gpurun_out/$TAG/$pn.log 2>&1 || { rc=$?; tail -5 gpurun_out/$TAG/$pn.log; break; }
  echo "$pn done"
done
kill $TICK
python3 tools/pmc_ntt.py gpurun_out/$TAG $CFG > gpurun_out/$TAG/summary.txt 2>&1
find gpurun_out/$TAG -name "*.csv" -size +20M -delete
cat gpurun_out/$TAG/summary.txt
exit $rc
