#!/bin/bash
# final tree: full GPU suite, smoke, the default bench line
set -o pipefail
timeout -k 10 1000 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gputests_final.log 2>&1 || { tail -30 gpurun_out/gputests_final.log; exit 1; }
tail -2 gpurun_out/gputests_final.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" || exit 1
timeout -k 10 900 python3 bench.py > gpurun_out/bench_p4_v4.log 2>&1 || { tail -5 gpurun_out/bench_p4_v4.log; exit 1; }
grep -a "^{" gpurun_out/bench_p4_v4.log | tail -1 > gpurun_out/bench_p4_v4.json
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/bench_p4_v4.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline_apply']['frac'])
print(d['cg'])
print(' '.join('%.3f' % r['apply_frac'] for r in d['by_order']))
for r in d['by_size']: print(r['problem'], r['dofs'], round(r['apply_frac'],3), round(r.get('cg_frac',0),3))
print(d['cxx_drivers']['rows']); print(d['cpu_baseline']['value'], d['cpu_baseline']['cg_dofs_iter_per_s'], d['setup_s'])
PY
