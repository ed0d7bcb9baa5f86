#!/bin/bash
# XCD runs by size: where does the threshold sit (p = 4, BP3: n = 78 (3.1e7 dofs), 92 (5e7), 116 (1e8)), and longer runs at 1e8; p = 2, 6 at 1e8
set -o pipefail
one() { # p n g
  MFEM_AMD_APPLY_XCD=$3 timeout -k 10 300 python -u bench.py --problem bp3 --p $1 --elems $2 --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 20 --steps 20 --warmup 5 > gpurun_out/j22.json 2>gpurun_out/j22.err || { tail -5 gpurun_out/j22.err; exit 1; }
  python3 -c "
import json
r=json.loads(open('gpurun_out/j22.json').read().strip().splitlines()[-1])
print('G=$3', 'bp3 p$1 n$2', r['config']['local_dofs_per_gpu'], 'apply ms', round(r['ms_per_step'],4), 'frac', round(r['roofline_apply']['frac'],4), 'l2e ms', round(r['roofline']['avg_launch_ms'],4), 'cg ms/it', round(r['cg']['ms_per_iter'],4), flush=True)"
}
for n in 78 92; do for g in 0 64; do one 4 $n $g; done; done
for g in 128 256 1024; do one 4 116 $g; done
for g in 0 64; do one 2 232 $g; one 6 77 $g; done
