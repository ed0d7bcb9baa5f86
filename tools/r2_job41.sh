#!/bin/bash
mkdir -p gpurun_out
examples/bin/vcoef_check -p 2 -n 3; examples/bin/vcoef_check -p 3 -n 2 -fn
timeout -k 10 900 python -u -m pytest tests/test_gpu_cxx.py tests/test_gpu_parity.py tests/test_gpu_bakeoff.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_part.log 2>&1 || { tail -30 gpurun_out/gpu_part.log; exit 1; }
tail -2 gpurun_out/gpu_part.log
timeout -k 10 300 python3 bench.py --problem bp5 --no-cpu --no-sizes --cg-iters 0 > gpurun_out/bench_bp5.log 2>&1 || exit 1
grep -a "^{" gpurun_out/bench_bp5.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('bp5 by_order us:', ' '.join('%.1f' % (r['ms_per_step']*1e3) for r in d['by_order']))"
