#!/bin/bash
for a in "-p 2 -n 3" "-p 2 -n 3 -fn" "-p 3 -n 2 -fn" "-p 4 -n 2 -fn" "-p 1 -n 3"; do examples/bin/vcoef_check $a | tail -1; done
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_gpu_cxx.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_part.log 2>&1 || { tail -30 gpurun_out/gpu_part.log; exit 1; }
tail -2 gpurun_out/gpu_part.log
timeout -k 10 300 python3 bench.py --problem bp5 --no-cpu --no-sizes --cg-iters 0 > gpurun_out/bench_bp5.log 2>&1 || exit 1
grep -a "^{" gpurun_out/bench_bp5.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('bp5 by_order us:', ' '.join('%.1f' % (r['ms_per_step']*1e3) for r in d['by_order']))"
