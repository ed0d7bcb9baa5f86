#!/bin/bash
# round 3, job 1: packed E-vector -- correctness (direct-rows tests, parity) then A/B against the holed layout
set -o pipefail
O=gpurun_out/r3job1; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_direct_rows.py tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
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
for bp in 3 1 5; do for p in 4 5 6; do
  for cfg in MFEM_AMD_PACKED=0 base MFEM_AMD_PACKED=0 base; do run $cfg $bp $p || exit 1; done
done; done
for cfg in 0 1; do
  MFEM_AMD_PACKED=$cfg timeout -k 10 500 python -u bench.py --no-sizes --no-cpu > $O/bench_packed$cfg.json 2>$O/bench_packed$cfg.err || { tail -5 $O/bench_packed$cfg.err; exit 1; }
  python3 -c "
import json,sys
r=json.loads(open('$O/bench_packed$cfg.json').read().strip().splitlines()[-1])
print('packed=$cfg', 'apply ms', round(r['ms_per_step'],4), 'frac', round(r['roofline_apply']['frac'],4), 'l2e ms', round(r['roofline']['avg_launch_ms'],4), 'e2l ms', round(r['roofline_apply']['e2l_avg_launch_ms'],4), 'cg', round(r['cg']['hbm_frac_per_gpu'],4))
print(' by_order', [round(x['apply_frac'],3) for x in r['by_order']])
" | tee -a $O/summary.txt
done
