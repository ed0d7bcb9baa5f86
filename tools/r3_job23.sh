#!/bin/bash
# the size-dependent XCD policy in place: 1e8 default vs forced round-robin; PCG at 1e7 with / without runs
set -o pipefail
one() { # tag p n env
  env $4 timeout -k 10 300 python -u bench.py --problem bp3 --p $2 --elems $3 --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 60 --steps 30 --warmup 5 > gpurun_out/j23.json 2>gpurun_out/j23.err || { tail -5 gpurun_out/j23.err; exit 1; }
  python3 -c "
import json
r=json.loads(open('gpurun_out/j23.json').read().strip().splitlines()[-1])
print('$1', 'bp3 p$2 n$3', r['config']['local_dofs_per_gpu'], 'apply ms', round(r['ms_per_step'],4), 'frac', round(r['roofline_apply']['frac'],4), 'l2e ms', round(r['roofline']['avg_launch_ms'],4), 'cg ms/it', round(r['cg']['ms_per_iter'],4), 'cg frac', round(r['cg']['hbm_frac_per_gpu'],4), flush=True)"
}
one default 4 116 X=0
one forced_rr 4 116 MFEM_AMD_APPLY_XCD=0
one default 4 54 X=0
one runs64 4 54 MFEM_AMD_APPLY_XCD=64
one runs16 4 54 MFEM_AMD_APPLY_XCD=16
one default 6 36 X=0
one runs64 6 36 MFEM_AMD_APPLY_XCD=64
one default 4 54 X=0
