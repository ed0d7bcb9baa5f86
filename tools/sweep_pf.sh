#!/bin/bash
# usage: tools/sweep_pf.sh "<problems>" "<orders>"  -- L->E kernel time for every (schedule mode, pa_data prefetch distance)
for pf in 0 2 3; do for m in 0 1 2; do echo "== PF $pf MODE $m"; MFEM_AMD_APPLY_PF=$pf MFEM_AMD_APPLY_MODE=$m timeout -k 10 300 python tools/kbench.py --problem $1 --p $2 --which l2e 2>&1 | grep -v amdgpu.ids | sed 's/| E2L.*//' || exit 1; done; done
