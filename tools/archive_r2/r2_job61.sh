#!/bin/bash
# Jacobi-PCG vector kernels: cached vs non-temporal streams for x, r, dinv (z, d cached) -- A/B/A/B in one job
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/job61; mkdir -p $O
for nt in 0 1 0 1; do
  for a in "bp3 4" "bp3 6" "bp1 4"; do
    echo "PCG_NT=$nt $a: $(MFEM_AMD_PCG_NT=$nt timeout -k 10 200 python -u tools/pcg_only.py $a 2>&1 | grep ms/iter)" | tee -a $O/summary.txt
  done
done
