#!/bin/bash
# p = 7, 8 (streamed pa_data, one or two waves per SIMD): ring depth PF = 2 (dispatched) against PF = 3
set -o pipefail
for pf in 2 3 2 3; do
for pp in "bp3 7" "bp3 8" "bp5 8" "bp1 8"; do set -- $pp
  MFEM_AMD_APPLY_PF=$pf timeout -k 10 200 python -u bench.py --problem $1 --p $2 --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 0 --steps 50 --warmup 10 > gpurun_out/j31.json 2>gpurun_out/j31.err || { echo "PF=$pf $1 p$2 failed"; tail -2 gpurun_out/j31.err; continue; }
  python3 -c "
import json
r=json.loads(open('gpurun_out/j31.json').read().strip().splitlines()[-1])
print('PF=$pf', '$1 p$2', 'apply ms', round(r['ms_per_step'],4), 'frac', round(r['roofline_apply']['frac'],4), 'l2e ms', round(r['roofline']['avg_launch_ms'],4), flush=True)"
done
done
