cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04streamprof; mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o p -- python3 $R/tools/bench_stream.py --snps 65536 --quick --dir $R > $O/log.txt 2>&1
cd $R; find $O -name "*kernel_trace.csv" -delete; rm -f sfg_stream_bench.bed
python3 - <<P
import csv,glob
f=glob.glob("gpurun_out/r04streamprof/prof/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:24]:
    print(f"{r['Name'][:60]:60s} calls {int(r['Calls']):7d} total_ms {float(r['TotalDurationNs'])/1e6:9.1f} avg_us {float(r['AverageNs'])/1e3:10.1f}")
P
tail -1 $O/log.txt | cut -c1-400
