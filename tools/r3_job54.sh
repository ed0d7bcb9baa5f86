#!/bin/bash
# transpose-gather + d.Ad kernel: the loop over the direct (element-interior) rows with four strides per trip (probe) against
# the plain loop (shipped): PCG ms per iteration, BP3 / BP1 / BP5 p = 4 and BP3 p = 6
one() { # lib prob p
  v=$(env MFEM_AMD_LIB=$PWD/mfem_amd/$1 timeout -k 10 120 python -u bench.py --problem $2 --p $3 --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 100 --steps 30 --warmup 10 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cg %.4f ms/it  %.4f  final_nom %r' % (r['cg']['ms_per_iter'], r['cg']['hbm_frac_per_gpu'], r['cg']['final_nom']))")
  echo "$2 p=$3 $1: $v"
}
for rep in 1 2; do
for pp in "bp3 4" "bp1 4" "bp5 4" "bp3 6"; do set -- $pp
one libmfem_amd.so $1 $2; one libprobe_diru4.so $1 $2
done
done
