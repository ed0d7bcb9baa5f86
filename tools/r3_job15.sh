#!/bin/bash
# Staged stores in every specialised kernel: parity first (direct rows, golden cases, bake-off), then the bench with the by-order table.
set -o pipefail
timeout -k 10 900 python -u -m pytest tests/test_gpu_direct_rows.py tests/test_gpu_parity.py tests/test_gpu_bakeoff.py tests/test_gpu_vector.py -x -q --timeout 300 --timeout-method thread > gpurun_out/j15_tests.log 2>&1 || { tail -30 gpurun_out/j15_tests.log; exit 1; }
tail -2 gpurun_out/j15_tests.log
timeout -k 10 500 python -u bench.py --no-cpu --no-sizes > gpurun_out/j15_bench.json 2>gpurun_out/j15_bench.err || { tail -5 gpurun_out/j15_bench.err; exit 1; }
python3 -c "
import json
r=json.loads(open('gpurun_out/j15_bench.json').read().strip().splitlines()[-1])
print('apply ms', round(r['ms_per_step'],4), 'frac', round(r['roofline_apply']['frac'],4), 'l2e ms', round(r['roofline']['avg_launch_ms'],4), 'e2l ms', round(r['roofline_apply']['e2l_avg_launch_ms'],4), 'cg', round(r['cg']['hbm_frac_per_gpu'],4))
print(' by_order', [round(x['apply_frac'],3) for x in r['by_order']])
print(' by_order ms', [round(x['ms_per_step'],4) for x in r['by_order']])
print(' cxx', [(x['p'], x['cxx_apply_frac'], x['cxx_pcg_frac']) for x in r['cxx_drivers']['rows']])
"
for prob in bp1 bp5; do
for p in 2 3 4 5 6 7 8; do
timeout -k 10 200 python -u bench.py --problem $prob --p $p --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 0 --steps 50 --warmup 10 > gpurun_out/j15.json 2>gpurun_out/j15.err || { tail -5 gpurun_out/j15.err; exit 1; }
python3 -c "
import json
r=json.loads(open('gpurun_out/j15.json').read().strip().splitlines()[-1])
print('$prob p$p', 'apply ms', round(r['ms_per_step'],4), 'frac', round(r['roofline_apply']['frac'],4), 'l2e ms', round(r['roofline']['avg_launch_ms'],4), 'e2l ms', round(r['roofline_apply']['e2l_avg_launch_ms'],4), flush=True)"
done
done
