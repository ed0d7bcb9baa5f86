#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 300 python3 tools/e2l_classmajor_probe.py > gpurun_out/e2l_classmajor_probe.txt 2>&1 || { tail -5 gpurun_out/e2l_classmajor_probe.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/e2l_classmajor_probe.txt
timeout -k 10 900 python -u -m pytest tests/test_gpu_cxx.py tests/test_gpu_parity.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_part.log 2>&1 || { tail -20 gpurun_out/gpu_part.log; exit 1; }
tail -2 gpurun_out/gpu_part.log
