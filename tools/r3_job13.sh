#!/bin/bash
# What does the store phase of the element kernel cost?  Probe build with the stores compiled out (values kept live), p = 4.
set -o pipefail
for lib in libmfem_amd.so libprobe_nostore.so libmfem_amd.so libprobe_nostore.so; do
  for prob in bp3 bp1; do
    MFEM_AMD_LIB=$PWD/mfem_amd/$lib timeout -k 10 200 python -u bench.py --problem $prob --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 0 --steps 50 --warmup 10 > gpurun_out/j13.json 2>gpurun_out/j13.err || { tail -5 gpurun_out/j13.err; exit 1; }
    python3 -c "
import json
r=json.loads(open('gpurun_out/j13.json').read().strip().splitlines()[-1])
print('$lib', '$prob', 'apply ms', round(r['ms_per_step'],4), 'frac', round(r['roofline_apply']['frac'],4), 'l2e ms', round(r['roofline']['avg_launch_ms'],4), 'e2l ms', round(r['roofline_apply']['e2l_avg_launch_ms'],4), flush=True)"
  done
done
