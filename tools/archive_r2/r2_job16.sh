#!/bin/bash
# the one-off stall inside a timed loop: same case sequence with and without a collection before timing
mkdir -p gpurun_out
KBENCH_GC=0 timeout -k 10 300 python3 tools/kbench.py --problem bp3,bp1,bp5 --p 1,2,3,4,5,6,7,8 --which l2e > gpurun_out/stall_gc0.txt 2>&1 || exit 1
KBENCH_GC=1 timeout -k 10 300 python3 tools/kbench.py --problem bp3,bp1,bp5 --p 1,2,3,4,5,6,7,8 --which l2e > gpurun_out/stall_gc1.txt 2>&1 || exit 1
paste <(awk '/L2E/{print $1,$3,$8}' gpurun_out/stall_gc0.txt) <(awk '/L2E/{print $8}' gpurun_out/stall_gc1.txt)
