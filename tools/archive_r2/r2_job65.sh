#!/bin/bash
# C++ driver (MFEM numbering), p = 2: which E->L form?
set -o pipefail
O=gpurun_out/job65; mkdir -p $O
for cfg in base MFEM_AMD_SCATTER_XCD=0 MFEM_AMD_SCATTER_BLK=0 base MFEM_AMD_SCATTER_XCD=0; do
  if [ "$cfg" = base ]; then envs=""; else envs="$cfg"; fi
  for bp in 3 1; do
    r=$(env $envs timeout -k 10 300 examples/bin/bench_ceed -bp $bp -p 2 -n 108 -mi 30 2>>$O/err.log | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['phase'], r.get('ms_per_apply', r.get('solve_s')), round(r['hbm_frac'],4), end='  ')
")
    echo "$cfg bp$bp: $r" | tee -a $O/summary.txt
  done
done
