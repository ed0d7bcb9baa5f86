#!/bin/bash
# elements per workgroup of the (5,6) kernels (NBZ for Q = 6: 7 shipped, 6, 5), in context: BP3 and BP1 p = 4
mkdir -p gpurun_out
for rep in 1 2; do for v in default NBZ6 NBZ5; do
  lib=$PWD/mfem_amd/libmfem_amd.so; [ $v != default ] && lib=$PWD/mfem_amd/libmfem_amd_probe$v.so
  for prob in bp3 bp1; do
  MFEM_AMD_LIB=$lib timeout -k 10 300 python3 bench.py --problem $prob --no-cpu --no-sizes --no-orders --cg-iters 20 > gpurun_out/bench_nbz.log 2>&1 || { tail -3 gpurun_out/bench_nbz.log; exit 1; }
  grep -a "^{" gpurun_out/bench_nbz.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$v $prob', 'p4 %.1f us' % (d['ms_per_step']*1e3), 'l2e %.1f' % (d['roofline']['avg_launch_ms']*1e3), 'cg %.1f' % (d['cg']['ms_per_iter']*1e3))"
  done
done; done
