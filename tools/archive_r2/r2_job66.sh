#!/bin/bash
# C++ drivers (the reference's numbering): do the cache / direct-row policies tuned on the synthetic boxes hold there?
set -o pipefail
O=gpurun_out/job66; mkdir -p $O
run() {  # cfg bp p
  cfg=$1; if [ "$cfg" = base ]; then envs=""; else envs="$cfg"; fi
  r=$(env $envs timeout -k 10 300 examples/bin/bench_ceed -bp $2 -p $3 -n $((216/$3)) -mi 50 2>>$O/err.log | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['phase'], round(r['ms_per_apply'],4) if 'ms_per_apply' in r else round(r['solve_s']/r['iterations']*1e3,4), round(r['hbm_frac'],4), end='  ')
")
  echo "$cfg bp$2 p$3: $r" | tee -a $O/summary.txt
}
for p in 2 4 6 8; do
  for cfg in base MFEM_AMD_DIRECT=0 MFEM_AMD_APPLY_NT=0 MFEM_AMD_SCATTER_NT=0 MFEM_AMD_PCG_NT=0 base; do run $cfg 3 $p || exit 1; done
done
for cfg in base MFEM_AMD_DIRECT=0 MFEM_AMD_APPLY_NT=0 MFEM_AMD_PCG_NT=0; do run $cfg 1 4 || exit 1; done
