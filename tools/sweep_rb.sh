#!/bin/bash
# usage: tools/sweep_rb.sh "<problems>" "<orders>" ["<rb levels>" ["<pf>"]] -- L->E kernel time per contraction variant
# (MFEM_AMD_APPLY_RB: 0 one output per thread and sweep, 1 row-per-thread x / y^T / x^T, 2 = 1 + row-per-thread y)
for pf in ${4:-default}; do for rb in ${3:-0 1 2}; do echo "== RB $rb PF $pf"; if [ $pf != default ]; then export MFEM_AMD_APPLY_PF=$pf; fi; MFEM_AMD_APPLY_RB=$rb timeout -k 10 400 python tools/kbench.py --problem $1 --p $2 --which l2e 2>&1 | grep -v amdgpu.ids | sed 's/| E2L.*//' || exit 1; done; done
