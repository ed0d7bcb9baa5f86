#!/bin/bash
# A/B on one box: staged stores (libmfem_amd.so) against direct per-row stores (libmfem_amd_nostage.so), every order and problem;
# then K batches per workgroup (MFEM_AMD_APPLY_BPW) with the staged form.
set -o pipefail
one() {  # lib prob p tag env...
  lib=$1; prob=$2; p=$3; tag=$4; shift 4
  env "$@" MFEM_AMD_LIB=$PWD/mfem_amd/$lib timeout -k 10 200 python -u bench.py --problem $prob --p $p --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 0 --steps 50 --warmup 10 > gpurun_out/j16.json 2>gpurun_out/j16.err || { echo "$tag $prob p$p FAILED"; tail -3 gpurun_out/j16.err; return 0; }
  python3 -c "
import json
r=json.loads(open('gpurun_out/j16.json').read().strip().splitlines()[-1])
print('$tag', '$prob p$p', 'apply ms', round(r['ms_per_step'],4), 'frac', round(r['roofline_apply']['frac'],4), 'l2e ms', round(r['roofline']['avg_launch_ms'],4), 'e2l ms', round(r['roofline_apply']['e2l_avg_launch_ms'],4), flush=True)"
}
for prob in bp3 bp1 bp5; do
  for p in 2 3 4 5 6 7 8; do
    one libmfem_amd_nostage.so $prob $p nostage X=0
    one libmfem_amd.so $prob $p stage X=0
  done
done
for k in 2 3 4; do
  for prob in bp3 bp1; do
    for p in 4 5 6; do
      one libmfem_amd.so $prob $p bpw$k MFEM_AMD_APPLY_BPW=$k
    done
  done
done
