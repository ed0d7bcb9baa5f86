#!/bin/bash
# 1e8 dofs: XCD-contiguous row blocks in the transpose-gather (MFEM_AMD_SCATTER_XCD=1)?
set -o pipefail
for prob in bp3 bp1; do
for f in 0 1; do
  MFEM_AMD_SCATTER_XCD=$f timeout -k 10 300 python -u bench.py --problem $prob --p 4 --elems 116 --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 20 --steps 20 --warmup 5 > gpurun_out/j24.json 2>gpurun_out/j24.err || { tail -5 gpurun_out/j24.err; exit 1; }
  python3 -c "
import json
r=json.loads(open('gpurun_out/j24.json').read().strip().splitlines()[-1])
print('scatter_xcd=$f', '$prob p4 n116', 'apply ms', round(r['ms_per_step'],4), 'frac', round(r['roofline_apply']['frac'],4), 'l2e ms', round(r['roofline']['avg_launch_ms'],4), 'e2l ms', round(r['roofline_apply']['e2l_avg_launch_ms'],4), 'cg ms/it', round(r['cg']['ms_per_iter'],4), flush=True)"
done
done
