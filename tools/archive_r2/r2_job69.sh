#!/bin/bash
# C++ drivers (reference numbering): E->L form at p = 1, 3, 4, 6
set -o pipefail
O=gpurun_out/job69; mkdir -p $O
run() {
  cfg=$1; if [ "$cfg" = base ]; then envs=""; else envs="$cfg"; fi
  r=$(env $envs timeout -k 10 300 examples/bin/bench_ceed -bp $2 -p $3 -n $((216/$3)) -mi 30 2>>$O/err.log | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['phase'], round(r['ms_per_apply'],4) if 'ms_per_apply' in r else round(r['solve_s']/r['iterations']*1e3,4), round(r['hbm_frac'],4), end='  ')
")
  echo "$cfg bp$2 p$3: $r" | tee -a $O/summary.txt
}
for p in 1 3 4 6; do
  for cfg in base MFEM_AMD_SCATTER_BLK=0 MFEM_AMD_SCATTER_XCD=1 base; do run $cfg 3 $p || exit 1; done
done
