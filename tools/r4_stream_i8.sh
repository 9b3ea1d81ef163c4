cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_stream_i8
for r in 80 24; do
SFG_I8_KEEP_RESERVE_GB=$r timeout -k 10 700 python3 tools/bench_stream.py --snps 65536 --dir $GRAFT_REPO_ROOT > gpurun_out/r04_stream_i8/log_$r.txt 2>&1; rc=$?
echo "reserve=$r rc=$rc"; tail -1 gpurun_out/r04_stream_i8/log_$r.txt | cut -c1-420
done
rm -f sfg_stream_bench.bed
