# per-kernel same-box A/B: bash tools/r2_kern_ab.sh <tag> <config> <lib> ...   (rocprofv3 kernel stats of one bench pass per library, single queue)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=$1; CFG=$2; shift 2; mkdir -p $R/gpurun_out/$TAG; cd $R
export SFG_MM_NO_OVERLAP=1
for lib in "$@"; do
  name=$(basename $lib .so)
  SFG_LIB_PATH=$R/$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/$name -o s -- python3 bench.py --config $CFG --no-cpu-baseline --no-check --steps 1 --warmup 0 > gpurun_out/$TAG/$name.log 2>&1 || { echo "FAILED $lib"; tail -5 gpurun_out/$TAG/$name.log; exit 1; }
  find gpurun_out/$TAG/$name -name "*kernel_trace.csv" -delete; find gpurun_out/$TAG/$name -name "*.db" -delete
  python3 - <<PY | tee -a gpurun_out/$TAG/ab.txt
import csv, json, glob
rows = {r['Name'].split('(')[0][:28]: r for r in csv.DictReader(open(glob.glob("gpurun_out/$TAG/$name/*kernel_stats.csv")[0]))}
line = [l for l in open("gpurun_out/$TAG/$name.log") if l.startswith('{')]
dig = json.loads(line[-1])['digests']['out1_sha256'][:12] if line else '?'
out = "%-28s" % "$name"
for k in ("void k_fft_encode<false>", "void k_ntt_half3<false, true", "k_ntt_half3", "void k_mac_bc<false, 30>", "k_ntt_fwd_split", "k_ntt_inv", "k_ksw_finish", "k_ksw_inner", "k_ksw_extend", "k_moddown_extend", "k_skew", "void k_skew<false>", "void k_skew<true>"):
    r = rows.get(k[:28]); out += ("  %s %8.1f us" % (k.split()[-1][:14], float(r['AverageNs']) / 1e3) if r else "")
print(out, dig)
PY
done
