# association scan: int8 rot tiles vs the fp64 rotation cache, fixed cost per call and marginal cost per batch separated
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_stream_ab
if [ -n "$TESTS" ]; then timeout -k 10 500 python -m pytest tests/test_gpu_stream.py tests/test_gpu_pgen.py -x -q -m gpu > gpurun_out/r04_stream_ab/tests.log 2>&1; rc=$?; tail -3 gpurun_out/r04_stream_ab/tests.log; [ $rc = 0 ] || exit $rc; fi
for v in 1 0; do
SFG_ASSOC_I8=$v timeout -k 10 500 python3 tools/bench_stream.py --snps ${SNPS:-65536} --dir $GRAFT_REPO_ROOT > gpurun_out/r04_stream_ab/stream_$v.txt 2>&1; rc=$?
echo "assoc_i8=$v rc=$rc"; tail -1 gpurun_out/r04_stream_ab/stream_$v.txt | cut -c1-900
done
rm -f sfg_stream_bench.bed sfg_stream_bench.bed.half
