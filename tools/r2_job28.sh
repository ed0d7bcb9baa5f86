#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_vector.py tests/test_gpu_bakeoff.py tests/test_gpu_matcoef.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_part.log 2>&1 || { tail -20 gpurun_out/gpu_part.log; exit 1; }
tail -2 gpurun_out/gpu_part.log
timeout -k 10 900 python3 bench.py > gpurun_out/bench_v4.log 2>&1 || { tail -5 gpurun_out/bench_v4.log; exit 1; }
grep -a "^{" gpurun_out/bench_v4.log | tail -1 > gpurun_out/bench_v4.json
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/bench_v4.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline_apply']['frac'])
print(d['cg'])
for r in d['by_order']: print(r['problem'], r['p'], round(r['apply_frac'],3), round(r['GDOF/s'],2), round(r['l2e_frac'],3))
for r in d['by_size']: print(r['problem'], r['dofs'], round(r['apply_frac'],3), round(r.get('cg_frac',0),3))
PY
