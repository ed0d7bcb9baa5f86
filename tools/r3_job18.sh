#!/bin/bash
# transpose-gather variants: parity (bit-exact E->L tests), then e2l times by order
set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_direct_rows.py tests/test_gpu_subrows.py -x -q --timeout 300 --timeout-method thread > gpurun_out/j18_tests.log 2>&1 || { tail -30 gpurun_out/j18_tests.log; exit 1; }
tail -1 gpurun_out/j18_tests.log
for prob in bp3 bp1; do
for p in 2 3 4 5 6 8; do
  timeout -k 10 200 python -u bench.py --problem $prob --p $p --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 30 --steps 50 --warmup 10 > gpurun_out/j18.json 2>gpurun_out/j18.err || { tail -5 gpurun_out/j18.err; exit 1; }
  python3 -c "
import json
r=json.loads(open('gpurun_out/j18.json').read().strip().splitlines()[-1])
print('$prob p$p', 'apply ms', round(r['ms_per_step'],4), 'frac', round(r['roofline_apply']['frac'],4), 'l2e ms', round(r['roofline']['avg_launch_ms'],4), 'e2l ms', round(r['roofline_apply']['e2l_avg_launch_ms'],4), 'cg ms/it', round(r['cg']['ms_per_iter'],4), 'final_nom', repr(r['cg']['final_nom']), flush=True)"
done
done
