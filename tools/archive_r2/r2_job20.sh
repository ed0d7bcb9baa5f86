#!/bin/bash
mkdir -p gpurun_out
bash tools/r2_job9.sh || exit 1
grep -a -o '"transport": "[^"]*"' gpurun_out/bench_g2_native_tcp.json gpurun_out/bench_g2_torch.json
grep -a -o '"value": [0-9.]*' gpurun_out/bench_g2_native_tcp.json gpurun_out/bench_g2_torch.json
timeout -k 10 600 python -u -m pytest tests/test_gpu_parallel.py tests/test_gpu_cxx.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_par.log 2>&1 || { tail -20 gpurun_out/gpu_par.log; exit 1; }
tail -2 gpurun_out/gpu_par.log
