#!/bin/bash
# round 3, job 2: packed E-vector (direct rows => packed), reference numbering in bench.py, self-launching --gpus N
set -o pipefail
O=gpurun_out/r3job2; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests/test_gpu_parallel.py tests/test_gpu_cxx.py -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
run() {
  cfg=$1; if [ "$cfg" = base ]; then envs="A=1"; else envs="$cfg"; fi
  r=$(env $envs timeout -k 10 300 examples/bin/bench_ceed -bp $2 -p $3 -n $((216/$3)) -reps 50 -mi 30 2>>$O/err.log | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['phase'], round(r['ms_per_apply'],4) if 'ms_per_apply' in r else round(r['solve_s']/r['iterations']*1e3,4), round(r['hbm_frac'],4), end='  ')
")
  echo "$cfg bp$2 p$3: $r" | tee -a $O/summary.txt
}
for bp in 3 1 5; do for p in 3 4 5 6 7 8; do
  for cfg in MFEM_AMD_DIRECT=0 base; do run $cfg $bp $p || exit 1; done
done; done
timeout -k 10 900 python -u bench.py --no-cpu > $O/bench.json 2>$O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 -c "
import json,sys
r=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('apply ms', round(r['ms_per_step'],4), 'frac', round(r['roofline_apply']['frac'],4), 'l2e ms', round(r['roofline']['avg_launch_ms'],4), 'e2l ms', round(r['roofline_apply']['e2l_avg_launch_ms'],4), 'cg', round(r['cg']['hbm_frac_per_gpu'],4), 'setup', r['setup_s'])
print(' by_order', [round(x['apply_frac'],3) for x in r['by_order']])
print(' by_size', [(x['problem'], x['dofs'], round(x['apply_frac'],3), round(x.get('cg_frac',0),3)) for x in r['by_size']])
print(' cxx', r['cxx_drivers'])
" | tee -a $O/summary.txt
