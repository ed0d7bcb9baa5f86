#!/bin/bash
mkdir -p gpurun_out
for nt in 0 1 0 1; do
  MFEM_AMD_SCATTER_NT=$nt timeout -k 10 300 python3 bench.py --no-cpu --no-sizes --cg-iters 20 > gpurun_out/bench_nt$nt.log 2>&1 || { tail -3 gpurun_out/bench_nt$nt.log; exit 1; }
  grep -a "^{" gpurun_out/bench_nt$nt.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('NT=$nt', 'p4 %.1f us' % (d['ms_per_step']*1e3), 'cg %.1f us' % (d['cg']['ms_per_iter']*1e3), ' by_order us:', ' '.join('%.1f' % (r['ms_per_step']*1e3) for r in d['by_order']))"
done
