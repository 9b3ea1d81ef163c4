# round 4: what does k_mac_i8 really fetch?  Three launches of the same operands (config c2: Q'*X^T runs K = 1183, the c4-sized contraction) -
# default (six column waves of a pair in one workgroup, rot tiles shared through the caches), SFG_MAC_I8_WG=1 (one wave per workgroup, the six on
# different XCDs: no sharing possible), SFG_MAC_I8_ROT=lds (rot tiles staged through LDS once per workgroup) - each under two counter passes:
#   P1  TCC_EA0_RDREQ_{32B,64B,128B}_sum + TCC_EA0_RDREQ_sum : exact fabric-side read bytes = 32 n32 + 64 n64 + 128 n128 (no "x2" correction needed)
#   P2  TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum    : L2 hit rate
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r04pmc}; CFG=${2:-c2}; mkdir -p $R/gpurun_out/$TAG; cd $R
( while sleep 45; do echo "tick $(date +%T)"; done ) & TICK=$!
export SFG_MM_NO_OVERLAP=1 SFG_UPLOAD_BLOCKING=1
rc=0
for v in ${VARIANTS:-default wg1 lds}; do
  case $v in default) unset SFG_MAC_I8_WG SFG_MAC_I8_ROT;; wg1) export SFG_MAC_I8_WG=1; unset SFG_MAC_I8_ROT;; lds) unset SFG_MAC_I8_WG; export SFG_MAC_I8_ROT=lds;; esac
  for p in "P1 TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum TCC_EA0_RDREQ_sum" "P2 TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum"; do
    set -- $p; pn=$1; shift
    d=gpurun_out/$TAG/${v}_$pn; rm -rf $d
    timeout -k 10 ${TMO:-400} rocprofv3 --pmc "$@" --output-format csv -d $d -o p -- python3 bench.py --config $CFG --steps 1 --warmup 0 --no-cpu-baseline --no-check --no-digest > gpurun_out/$TAG/${v}_$pn.log 2>&1 || { rc=$?; tail -5 gpurun_out/$TAG/${v}_$pn.log; break 2; }
    echo "$v $pn done"
  done
done
kill $TICK
python3 tools/pmc_mac_i8.py gpurun_out/$TAG > gpurun_out/$TAG/summary.txt 2>&1
find gpurun_out/$TAG -name "*counter_collection.csv" -size +20M -delete
cat gpurun_out/$TAG/summary.txt
exit $rc
