#!/bin/bash
# staged stores at D = 9 where the table has them off: (9,9) diffusion, mass (9,10)
one() { # lib prob p label
  v=$(env MFEM_AMD_LIB=$PWD/mfem_amd/$1 timeout -k 10 120 python -u bench.py --problem $2 --p $3 --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 20 --steps 50 --warmup 10 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%7.1f us  %.4f  kernel %.4f  final_nom %r' % (r['ms_per_step']*1e3, r['roofline_apply']['frac'], r['roofline']['frac'], r['cg']['final_nom']))")
  echo "$2 p=$3 $4: $v"
}
for rep in 1 2; do
one libmfem_amd.so bp5 8 "table (direct)"; one libprobe_d9stg.so bp5 8 "staged"
one libmfem_amd.so bp1 8 "table (direct)"; one libprobe_d9stg.so bp1 8 "staged"
done
