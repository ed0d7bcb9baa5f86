#!/bin/bash
# round 3, job 6: grid-barrier cost probe; full GPU suite; bench.py with every leg
set -o pipefail
O=gpurun_out/r3job6; mkdir -p $O
for k in 1 2 4; do timeout -k 10 60 tools/gridbar_probe $k 2000 >> $O/gridbar_probe.txt 2>&1 || { cat $O/gridbar_probe.txt; exit 1; }; done
cat $O/gridbar_probe.txt
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 600 --timeout-method thread > $O/gpu_suite.log 2>&1 || { tail -40 $O/gpu_suite.log; exit 1; }
tail -3 $O/gpu_suite.log
timeout -k 10 900 python -u bench.py > $O/bench.json 2>$O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 -c "
import json,sys
r=json.loads(open('$O/bench.json').read().strip().splitlines()[-1])
print('apply ms', round(r['ms_per_step'],4), 'frac', round(r['roofline_apply']['frac'],4), 'l2e ms', round(r['roofline']['avg_launch_ms'],4), 'roofline.frac', round(r['roofline']['frac'],4), 'e2l ms', round(r['roofline_apply']['e2l_avg_launch_ms'],4), 'cg', round(r['cg']['hbm_frac_per_gpu'],4), 'setup', round(r['setup_s'],2))
print(' by_order', [round(x['apply_frac'],3) for x in r['by_order']])
print(' by_size', [(x['problem'], x['dofs'], round(x['apply_frac'],3), round(x.get('cg_frac',0),3)) for x in r['by_size']])
print(' cxx', [(x['p'], x.get('cxx_apply_frac'), x.get('cxx_pcg_frac')) for x in r['cxx_drivers']['rows']])
print(' cpu', r['cpu_baseline'])
" | tee -a $O/summary.txt
