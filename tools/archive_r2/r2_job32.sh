#!/bin/bash
mkdir -p gpurun_out
for v in 0 1 0 1; do
  MFEM_AMD_APPLY_NT=$v timeout -k 10 300 python3 bench.py --no-cpu --no-sizes --cg-iters 50 > gpurun_out/bench_ant.log 2>&1 || { tail -3 gpurun_out/bench_ant.log; exit 1; }
  grep -a "^{" gpurun_out/bench_ant.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('APPLY_NT=$v', 'p4 %.1f us' % (d['ms_per_step']*1e3), 'l2e %.1f' % (d['roofline']['avg_launch_ms']*1e3), 'cg %.1f us' % (d['cg']['ms_per_iter']*1e3), ' by_order us:', ' '.join('%.1f' % (r['ms_per_step']*1e3) for r in d['by_order']))"
done
