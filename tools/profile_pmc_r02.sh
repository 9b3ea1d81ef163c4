cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/r02p
# rocprofv3 --pmc serialises dispatches and misbehaves (hang / SIGSEGV inside the runtime) with the two-queue overlap and the
# stream-ordered pinned uploads at this size: the counter passes use the single-queue, blocking-upload path. Traffic per MAC launch is unaffected.
export SFG_MM_NO_OVERLAP=1 SFG_UPLOAD_BLOCKING=1
rm -rf gpurun_out/r02p/pmc_fetch gpurun_out/r02p/pmc_write
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/r02p/pmc_fetch -o f -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-check --no-digest > gpurun_out/r02p/pmc_fetch.log 2>&1
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/r02p/pmc_write -o w -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-check --no-digest > gpurun_out/r02p/pmc_write.log 2>&1
python3 tools/pmc_traffic.py gpurun_out/r02p/pmc_fetch gpurun_out/r02p/pmc_write gpurun_out/r02p/traffic.json > gpurun_out/r02p/traffic.txt 2>&1
find gpurun_out/r02p -name "*counter_collection.csv" -delete
head -4 gpurun_out/r02p/traffic.txt
