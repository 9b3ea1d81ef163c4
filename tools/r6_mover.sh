#!/bin/bash
# gpurun -- bash tools/r6_mover.sh  -> gpurun_out/r6_mover.txt
mkdir -p gpurun_out
timeout -k 10 900 python tools/r6_mover_ubench.py 13 > gpurun_out/r6_mover.txt 2>&1
echo "exit $?" >> gpurun_out/r6_mover.txt
tail -5 gpurun_out/r6_mover.txt
