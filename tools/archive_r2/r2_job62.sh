#!/bin/bash
# with the streaming policy on: do direct interior rows pay inside the solver now?
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/job62; mkdir -p $O
for dr in 1 2 1 2; do
  for a in "bp3 4" "bp3 6" "bp3 8" "bp1 4"; do
    echo "DIRECT=$dr $a: $(MFEM_AMD_DIRECT=$dr timeout -k 10 200 python -u tools/pcg_only.py $a 2>&1 | grep ms/iter)" | tee -a $O/summary.txt
  done
done
