#!/bin/bash
# (5,6) diffusion kernel at FOUR workgroups per CU: streamed pa_data ring (PF 2 / 3 / 4 slices, <= 124 VGPRs) + in-place
# contraction (32 KB of LDS) in the one-batch form, against the shipped whole-column kernel (143 VGPRs, three workgroups)
one() { # lib label env...
  lib=$1; label=$2; shift; shift
  v=$(env MFEM_AMD_LIB=$PWD/mfem_amd/$lib "$@" timeout -k 10 120 python -u bench.py --problem bp3 --p 4 --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 20 --steps 50 --warmup 10 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%7.1f us  %.4f  kernel %.4f  final_nom %r' % (r['ms_per_step']*1e3, r['roofline_apply']['frac'], r['roofline']['frac'], r['cg']['final_nom']))")
  echo "bp3 p=4 $label: $v"
}
for rep in 1 2; do
one libmfem_amd.so "shipped (whole columns)" X=0
one libprobe_d5deep.so "in place, whole columns" X=0
for pf in 2 3 4; do one libprobe_d5deep.so "in place, ring of $pf" MFEM_AMD_APPLY_PF=$pf; done
done
