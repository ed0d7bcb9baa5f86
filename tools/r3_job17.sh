#!/bin/bash
# What the transpose-gather's time is made of: no value gather at all (1), a gather that always hits (2), the kernel (0).  p = 4, 2, 6.
set -o pipefail
for lib in libmfem_amd.so libprobe_e2l1.so libprobe_e2l2.so libmfem_amd.so; do
  for p in 4 2 6; do
    MFEM_AMD_LIB=$PWD/mfem_amd/$lib timeout -k 10 200 python -u bench.py --p $p --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 0 --steps 50 --warmup 10 > gpurun_out/j17.json 2>gpurun_out/j17.err || { tail -5 gpurun_out/j17.err; exit 1; }
    python3 -c "
import json
r=json.loads(open('gpurun_out/j17.json').read().strip().splitlines()[-1])
print('$lib', 'bp3 p$p', 'apply ms', round(r['ms_per_step'],4), 'l2e ms', round(r['roofline']['avg_launch_ms'],4), 'e2l ms', round(r['roofline_apply']['e2l_avg_launch_ms'],4), flush=True)"
  done
done
