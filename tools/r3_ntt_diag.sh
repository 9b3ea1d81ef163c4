# timing diagnostics of the panel NTT (alt builds with -DSFG_NTT_DIAG, results invalid): per-kernel averages of one c2 pass, single queue
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; TAG=${1:-r03_nttdiag2}; shift; mkdir -p $R/gpurun_out/$TAG; cd $R
export SFG_MM_NO_OVERLAP=1
for lib in "$@"; do
  name=$(basename $lib .so)
  SFG_LIB_PATH=$R/$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG/$name -o s -- python3 bench.py --config c2 --no-cpu-baseline --no-check --no-digest --steps 1 --warmup 0 > gpurun_out/$TAG/$name.log 2>&1 || { echo "FAILED $lib"; tail -5 gpurun_out/$TAG/$name.log; exit 1; }
  echo "== $name"; grep -E "k_ntt_half3|k_fft_encode" $(find gpurun_out/$TAG/$name -name '*kernel_stats.csv') | awk -F'",' '{print substr($1,1,60), $2}' | cut -c1-120
  rm -rf gpurun_out/$TAG/$name
done
