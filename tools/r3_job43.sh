#!/bin/bash
# (5,6) diffusion kernel: 16-byte lanes in the staged copy-out (probe) against the shipped 8-byte form, now that the one-trip
# kernel has registers to spare
set -o pipefail
for rep in 1 2 3; do
for lib in libmfem_amd.so libprobe_d5w16.so; do
  MFEM_AMD_LIB=$PWD/mfem_amd/$lib timeout -k 10 200 python -u bench.py --problem bp3 --p 4 --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 30 --steps 50 --warmup 10 > gpurun_out/j43.json 2>gpurun_out/j43.err || { tail -5 gpurun_out/j43.err; exit 1; }
  python3 -c "
import json
r=json.loads(open('gpurun_out/j43.json').read().strip().splitlines()[-1])
print('$lib', 'bp3 p4', 'apply ms', round(r['ms_per_step'],4), 'frac', round(r['roofline_apply']['frac'],4), 'l2e ms', round(r['roofline']['avg_launch_ms'],4), 'final_nom', repr(r['cg']['final_nom']), flush=True)"
done
done
