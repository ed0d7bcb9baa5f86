#!/bin/bash
mkdir -p gpurun_out
bash tools/gpu_suite.sh || exit 1
bash tools/r2_job9.sh || exit 1
grep -a -o '"transport": "[^"]*"' gpurun_out/bench_g2_native_tcp.json gpurun_out/bench_g2_torch.json
grep -a -o '"value": [0-9.]*' gpurun_out/bench_g2_native_tcp.json gpurun_out/bench_g2_torch.json
