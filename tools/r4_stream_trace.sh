cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_stream_trace
timeout -k 10 500 python -m pytest tests/test_gpu_stream.py tests/test_gpu_pgen.py -x -q -m gpu > gpurun_out/r04_stream_trace/tests.log 2>&1; rc=$?; tail -3 gpurun_out/r04_stream_trace/tests.log; [ $rc = 0 ] || exit $rc
SFG_ASSOC_TRACE=1 timeout -k 10 500 python3 tools/bench_stream.py --snps ${SNPS:-65536} --dir $GRAFT_REPO_ROOT > gpurun_out/r04_stream_trace/stream_1.txt 2> gpurun_out/r04_stream_trace/trace_1.txt; rc=$?
echo "traced rc=$rc"; grep -v "batch [1-9]" gpurun_out/r04_stream_trace/trace_1.txt | sed -n 1,40p
timeout -k 10 500 python3 tools/bench_stream.py --snps ${SNPS:-65536} --dir $GRAFT_REPO_ROOT > gpurun_out/r04_stream_trace/stream_u.txt 2>&1; rc=$?
echo "untraced rc=$rc"; tail -1 gpurun_out/r04_stream_trace/stream_u.txt | cut -c1-700
rm -f sfg_stream_bench.bed sfg_stream_bench.bed.half
