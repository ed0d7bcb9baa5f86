#!/bin/bash
# D = 3 diffusion kernels: the split pa_data request (x staged between the two halves; adopted for the looping form) against
# the plain order on the one-trip kernels
one() { # lib prob p label
  v=$(env MFEM_AMD_LIB=$PWD/mfem_amd/$1 timeout -k 10 120 python -u bench.py --problem $2 --p $3 --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 20 --steps 50 --warmup 10 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%7.1f us  %.4f  kernel %.4f  final_nom %r' % (r['ms_per_step']*1e3, r['roofline_apply']['frac'], r['roofline']['frac'], r['cg']['final_nom']))")
  echo "$2 p=$3 $4: $v"
}
for rep in 1 2; do
one libmfem_amd.so bp3 2 "split"; one libprobe_d3nosplit.so bp3 2 "plain"
one libmfem_amd.so bp5 2 "split"; one libprobe_d3nosplit.so bp5 2 "plain"
done
