#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python3 bench.py > gpurun_out/bench_v5.log 2>&1 || { tail -5 gpurun_out/bench_v5.log; exit 1; }
grep -a "^{" gpurun_out/bench_v5.log | tail -1 > gpurun_out/bench_v5.json
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/bench_v5.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline_apply']['frac'])
print(d['cg'])
print(' '.join('%.3f' % r['apply_frac'] for r in d['by_order']))
for r in d['by_size']: print(r['problem'], r['dofs'], round(r['apply_frac'],3), round(r.get('cg_frac',0),3))
print(d['cpu_baseline'])
PY
