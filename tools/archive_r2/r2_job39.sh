#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1000 python3 tools/ctx_sweep.py bp3 4,5,6,7,8 > gpurun_out/ctx_sweep_bp3.txt 2>&1 || { tail -5 gpurun_out/ctx_sweep_bp3.txt; exit 1; }
cat gpurun_out/ctx_sweep_bp3.txt
