#!/bin/bash
# gather-map loads non-temporal (GM) against the library (pa_data non-temporal), in context
mkdir -p gpurun_out
for prob in bp3 bp1; do
for v in default GM default GM; do
  lib=$PWD/mfem_amd/libmfem_amd.so; [ $v != default ] && lib=$PWD/mfem_amd/libmfem_amd_probe$v.so
  MFEM_AMD_LIB=$lib timeout -k 10 300 python3 bench.py --problem $prob --no-cpu --no-sizes --cg-iters 0 > gpurun_out/bench_pant.log 2>&1 || { tail -3 gpurun_out/bench_pant.log; exit 1; }
  grep -a "^{" gpurun_out/bench_pant.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('$prob $v'.ljust(14), ' by_order us:', ' '.join('%7.1f' % (r['ms_per_step']*1e3) for r in d['by_order']))"
done; done
