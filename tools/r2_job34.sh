#!/bin/bash
mkdir -p gpurun_out
for v in 0 1 0 1; do
  MFEM_AMD_SCATTER_NT=$v timeout -k 10 300 python3 bench.py --no-cpu --no-sizes --no-orders --cg-iters 100 > gpurun_out/bench_snt.log 2>&1 || { tail -3 gpurun_out/bench_snt.log; exit 1; }
  grep -a "^{" gpurun_out/bench_snt.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('SCATTER_NT=$v', 'p4 %.1f us' % (d['ms_per_step']*1e3), 'cg %.1f us' % (d['cg']['ms_per_iter']*1e3))"
done
