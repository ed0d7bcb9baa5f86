#!/bin/bash
# Does the x re-fetch of the element kernel (different XCDs fetch the same L-vector lines) show in the memory-side counters,
# and does removing it (runs of 16 batches per XCD) change the time?  PMC passes with MFEM_AMD_APPLY_XCD=0 / 16, BP1 and BP3.
set -o pipefail
for g in 0 16; do
  for prob in bp1 bp3; do
    export MFEM_AMD_APPLY_XCD=$g
    bash tools/pmc_bench.sh xcdtraffic_${prob}_g$g probe_${prob}_g$g --problem $prob || exit 1
    timeout -k 10 120 python3 bench.py --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 0 --steps 50 --warmup 10 --problem $prob > gpurun_out/xcdtraffic_${prob}_g$g.json || exit 1
  done
done
