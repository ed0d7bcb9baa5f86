#!/bin/bash
# pa_data loads non-temporal everywhere (NT1) / nowhere (NT0) / per the table (default), timed IN CONTEXT
# (bench.py by_order: the alternating apply kernels), BP3 then BP1 and BP5 at all orders
mkdir -p gpurun_out
for prob in bp3 bp1 bp5; do
for v in default NT0 NT1; do
  lib=$PWD/mfem_amd/libmfem_amd.so; [ $v != default ] && lib=$PWD/mfem_amd/libmfem_amd_probe$v.so
  MFEM_AMD_LIB=$lib timeout -k 10 300 python3 bench.py --problem $prob --no-cpu --no-sizes --cg-iters 0 > gpurun_out/bench_pant.log 2>&1 || { tail -3 gpurun_out/bench_pant.log; exit 1; }
  grep -a "^{" gpurun_out/bench_pant.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$prob $v'.ljust(14), ' by_order us:', ' '.join('%7.1f' % (r['ms_per_step']*1e3) for r in d['by_order']))"
done; done
