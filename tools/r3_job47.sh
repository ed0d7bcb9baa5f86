#!/bin/bash
# Q = 3 kernels, elements per workgroup: 16 (shipped, 144 threads), 14 (126 threads = 2 waves), 7 (63 threads = ONE wave: the
# phase barriers become wave-level fences)
one() { # lib prob p label
  v=$(env MFEM_AMD_LIB=$PWD/mfem_amd/$1 timeout -k 10 120 python -u bench.py --problem $2 --p $3 --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 20 --steps 40 --warmup 10 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%7.1f us  %.4f  kernel %.4f  final_nom %r' % (r['ms_per_step']*1e3, r['roofline_apply']['frac'], r['roofline']['frac'], r['cg']['final_nom']))")
  echo "$2 p=$3 $4: $v"
}
for rep in 1 2; do
for prob in bp3 bp1; do
one libmfem_amd.so $prob 1 "nbz 16"; one libprobe_nbz3_14.so $prob 1 "nbz 14"; one libprobe_nbz3_7.so $prob 1 "nbz 7 "
done
done
