#!/bin/bash
# Q = 3 kernels ((2,3): BP3 p = 1, BP1 p = 1; (3,3): BP5 p = 2) with 28 elements per workgroup (252 threads) instead of 16 (144)
one() { # lib prob p label
  v=$(env MFEM_AMD_LIB=$PWD/mfem_amd/$1 timeout -k 10 120 python -u bench.py --problem $2 --p $3 --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 20 --steps 40 --warmup 10 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%7.1f us  %.4f  kernel %.4f  final_nom %r' % (r['ms_per_step']*1e3, r['roofline_apply']['frac'], r['roofline']['frac'], r['cg']['final_nom']))")
  echo "$2 p=$3 $4: $v"
}
for rep in 1 2; do
one libmfem_amd.so bp3 1 "nbz 16"; one libprobe_nbz3_d2.so bp3 1 "nbz 28"
one libmfem_amd.so bp1 1 "nbz 16"; one libprobe_nbz3_d2.so bp1 1 "nbz 28"
one libmfem_amd.so bp5 2 "nbz 16"; one libprobe_nbz3_d3.so bp5 2 "nbz 28"
done
