#!/bin/bash
# BP5 p = 4, 5 ((5,5), (6,6): RB = 2 in the table): RB = 1 with and without the in-place transposed y contraction
set -o pipefail
one() {  # lib problem p env...
  lib=$1; prob=$2; p=$3; shift; shift; shift
  env MFEM_AMD_LIB=$PWD/mfem_amd/$lib "$@" timeout -k 10 300 python -u bench.py --problem $prob --p $p --no-cpu --no-cxx --no-sizes --no-orders --cg-iters 30 --steps 50 --warmup 10 > gpurun_out/j39.json 2>gpurun_out/j39.err || { tail -5 gpurun_out/j39.err; exit 1; }
  python3 -c "
import json
r=json.loads(open('gpurun_out/j39.json').read().strip().splitlines()[-1])
print('$lib', '$prob', 'p$p', '$*', 'apply ms', round(r['ms_per_step'],4), 'frac', round(r['roofline_apply']['frac'],4), 'l2e', round(r['roofline']['frac'],4), 'cg ms/it', round(r['cg']['ms_per_iter'],4), 'final_nom', repr(r['cg']['final_nom']), flush=True)"
}
for rep in 1 2; do
for p in 4 5; do
one libmfem_amd.so bp5 $p X=0 && one libmfem_amd.so bp5 $p MFEM_AMD_APPLY_RB=1 && one libprobe_inpl.so bp5 $p MFEM_AMD_APPLY_RB=1 || exit 1
done
done
