#!/bin/bash
mkdir -p gpurun_out
for prob in bp1 bp5; do
timeout -k 10 500 python3 tools/ctx_sweep.py $prob 3,4,5,6,8 > gpurun_out/ctx_sweep_$prob.txt 2>&1 || { tail -5 gpurun_out/ctx_sweep_$prob.txt; exit 1; }
cat gpurun_out/ctx_sweep_$prob.txt
done
