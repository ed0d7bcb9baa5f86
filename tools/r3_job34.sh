#!/bin/bash
# D = 8 kernels with the new table (whole columns, one batch per workgroup, RB 1): staged stores on / off, p = 7, BP3 / BP5 / BP1; then the by-order lines
set -o pipefail
for rep in 1 2; do
for lib in libprobe_d8stage0.so libprobe_d8stage1.so; do
for prob in bp3 bp5 bp1; do
  MFEM_AMD_LIB=$PWD/mfem_amd/$lib timeout -k 10 200 python -u bench.py --problem $prob --p 7 --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 30 --steps 50 --warmup 10 > gpurun_out/j34.json 2>gpurun_out/j34.err || { tail -5 gpurun_out/j34.err; exit 1; }
  python3 -c "
import json
r=json.loads(open('gpurun_out/j34.json').read().strip().splitlines()[-1])
print('$lib', '$prob p7', 'apply ms', round(r['ms_per_step'],4), 'frac', round(r['roofline_apply']['frac'],4), 'l2e ms', round(r['roofline']['avg_launch_ms'],4), 'cg ms/it', round(r['cg']['ms_per_iter'],4), 'final_nom', repr(r['cg']['final_nom']), flush=True)"
done
done
done
