# (1) piece-gather / scatter probe, (2) association-scan tests + streamed bench on the int8 rot tiles vs the fp64 cache
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_gather
timeout -k 10 500 python -m pytest tests/test_gpu_stream.py tests/test_gpu_pgen.py -x -q -m gpu > gpurun_out/r04_gather/tests.log 2>&1; rc=$?; tail -3 gpurun_out/r04_gather/tests.log; [ $rc = 0 ] || exit $rc
for v in 1 0; do
SFG_ASSOC_I8=$v timeout -k 10 500 python3 tools/bench_stream.py --snps 65536 --dir $GRAFT_REPO_ROOT > gpurun_out/r04_gather/stream_$v.txt 2>&1; rc=$?
echo "assoc_i8=$v rc=$rc"; tail -1 gpurun_out/r04_gather/stream_$v.txt | cut -c1-600
done
rm -f sfg_stream_bench.bed
