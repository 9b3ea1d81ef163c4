#!/bin/bash
# tools/r5_streams.py over the stream counts / modes; one process each
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05streams.txt; : > $O
run() { python3 tools/r5_streams.py "$@" 2>&1 | grep -v "^\s*$" | tail -1 >> $O; }
run 0 unused
for n in 1 2 3 4 6; do run $n unused; done
run 1 used; run 2 used; run 4 used
run 1 before; run 2 before
run 1 on_extra; run 2 on_extra; run 4 on_extra
run 2 on_first; run 2 waits; run 4 waits
GPU_MAX_HW_QUEUES=2 run 2 unused
GPU_MAX_HW_QUEUES=2 run 2 used
GPU_MAX_HW_QUEUES=8 run 2 unused
GPU_MAX_HW_QUEUES=8 run 2 used
GPU_MAX_HW_QUEUES=1 run 2 used
run 0 unused
cat $O
