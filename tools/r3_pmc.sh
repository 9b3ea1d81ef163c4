# round 3: PMC HBM traffic passes of the default bench command (single queue + blocking uploads, as the counter mode needs); progress ticks keep the box's silence detector quiet
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${1:-r03p}; CFG=${2:-c4}; mkdir -p $R/gpurun_out/$TAG; cd $R
( while sleep 45; do echo "tick $(date +%T)"; done ) & TICK=$!
export SFG_MM_NO_OVERLAP=1 SFG_UPLOAD_BLOCKING=1
rc=0
for c in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/$TAG/pmc_$c; rm -rf $d
  timeout -k 10 ${TMO:-900} rocprofv3 --pmc $c --output-format csv -d $d -o p -- python3 bench.py --config $CFG --steps 1 --warmup 0 --no-cpu-baseline --no-check --no-digest > gpurun_out/$TAG/pmc_$c.log 2>&1 || { rc=$?; tail -5 gpurun_out/$TAG/pmc_$c.log; break; }
  grep -c . $d/*/*counter_collection.csv 2>/dev/null | head -2
done
kill $TICK
[ $rc = 0 ] && python3 tools/pmc_traffic.py gpurun_out/$TAG/pmc_FETCH_SIZE gpurun_out/$TAG/pmc_WRITE_SIZE gpurun_out/$TAG/traffic.json > gpurun_out/$TAG/traffic.txt 2>&1
find gpurun_out/$TAG -name "*counter_collection.csv" -delete
head -4 gpurun_out/$TAG/traffic.txt
exit $rc
