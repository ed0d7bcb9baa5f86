#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_bakeoff.py -m gpu -x -q --timeout 300 --timeout-method thread -k "cache_policy" > gpurun_out/gpu_part.log 2>&1 || { tail -30 gpurun_out/gpu_part.log; exit 1; }
tail -2 gpurun_out/gpu_part.log
