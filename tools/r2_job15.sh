#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python3 bench.py > gpurun_out/bench_v3.log 2>&1 || { tail -5 gpurun_out/bench_v3.log; exit 1; }
grep -a "^{" gpurun_out/bench_v3.log | tail -1 > gpurun_out/bench_v3.json
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/bench_v3.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_apply'])
print(d['cg'])
for r in d['by_order']: print(r['problem'], r['p'], round(r['apply_frac'],3), round(r['GDOF/s'],2))
for r in d['by_size']: print({k:(round(v,3) if isinstance(v,float) else v) for k,v in r.items()})
print(d['cpu_baseline'])
PY
