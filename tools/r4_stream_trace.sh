cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_stream_trace
for v in 1 0; do
SFG_ASSOC_TRACE=1 SFG_ASSOC_I8=$v timeout -k 10 500 python3 tools/bench_stream.py --snps ${SNPS:-65536} --dir $GRAFT_REPO_ROOT > gpurun_out/r04_stream_trace/stream_$v.txt 2> gpurun_out/r04_stream_trace/trace_$v.txt; rc=$?
echo "assoc_i8=$v rc=$rc"; grep -c batch gpurun_out/r04_stream_trace/trace_$v.txt; sed -n 1,60p gpurun_out/r04_stream_trace/trace_$v.txt
done
rm -f sfg_stream_bench.bed sfg_stream_bench.bed.half
