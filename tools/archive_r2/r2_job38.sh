#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_bakeoff.py tests/test_gpu_vector.py tests/test_gpu_vector_ref.py tests/test_gpu_matcoef.py tests/test_gpu_cxx.py tests/test_gpu_parallel.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_pcg.log 2>&1 || { tail -30 gpurun_out/gpu_pcg.log; exit 1; }
tail -2 gpurun_out/gpu_pcg.log
for v in 1 0 1 0; do
  MFEM_AMD_PCG_SIX=$v timeout -k 10 300 python3 bench.py --no-cpu --no-orders --cg-iters 100 > gpurun_out/bench_six.log 2>&1 || { tail -3 gpurun_out/bench_six.log; exit 1; }
  grep -a "^{" gpurun_out/bench_six.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('SIX=$v', 'cg %.1f us %.3f' % (d['cg']['ms_per_iter']*1e3, d['cg']['hbm_frac_per_gpu']), 'final_nom %.6e' % d['cg']['final_nom'], ' by_size cg_frac:', ' '.join('%.3f' % r['cg_frac'] for r in d['by_size']))"
done
