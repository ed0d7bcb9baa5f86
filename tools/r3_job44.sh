#!/bin/bash
# D = 9 with the in-place transposed y contraction forced on (probe, RB = 1, one batch per workgroup, whole columns) against the table
one() { # lib prob label env...
  lib=$1; prob=$2; label=$3; shift; shift; shift
  v=$(env MFEM_AMD_LIB=$PWD/mfem_amd/$lib "$@" timeout -k 10 120 python -u bench.py --problem $prob --p 8 --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 0 --steps 40 --warmup 10 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%6.1f us  %.4f' % (r['ms_per_step']*1e3, r['roofline_apply']['frac']))")
  echo "$prob p=8 $label: $v"
}
for rep in 1 2; do
for prob in bp5 bp3; do
one libmfem_amd.so $prob table X=0
one libmfem_amd.so $prob "mode 0 pf 0 rb 1" MFEM_AMD_APPLY_MODE=0 MFEM_AMD_APPLY_PF=0 MFEM_AMD_APPLY_RB=1
one libprobe_d9inpl.so $prob "mode 0 pf 0 rb 1 in place" MFEM_AMD_APPLY_MODE=0 MFEM_AMD_APPLY_PF=0 MFEM_AMD_APPLY_RB=1
done
done
