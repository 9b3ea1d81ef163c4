cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04_stream_g
timeout -k 10 500 python -m pytest tests/test_gpu_stream.py "tests/test_gpu_fullsize.py::test_c5_streamed_bed_batches_on_the_int8_rot_tiles_equal_the_resident_products" -x -q -m gpu > gpurun_out/r04_stream_g/tests.log 2>&1; rc=$?; tail -3 gpurun_out/r04_stream_g/tests.log; [ $rc = 0 ] || exit $rc
for g in auto; do
if [ $g = auto ]; then unset SFG_MM_GROUP; else export SFG_MM_GROUP=$g; fi
timeout -k 10 400 python3 tools/bench_stream.py --snps 65536 --dir $GRAFT_REPO_ROOT > gpurun_out/r04_stream_g/g$g.txt 2>&1; rc=$?
echo "G=$g rc=$rc"; tail -1 gpurun_out/r04_stream_g/g$g.txt | cut -c1-330
done
rm -f sfg_stream_bench.bed sfg_stream_bench.bed.half
