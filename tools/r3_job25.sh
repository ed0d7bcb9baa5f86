#!/bin/bash
# Coded gather map: parity, then A/B in one library (MFEM_AMD_GATHER_CODE=0 ignores the tables), every order from p = 4
set -o pipefail
timeout -k 10 600 python -u -m pytest tests/test_gpu_gather_code.py -x -q --timeout 300 --timeout-method thread > gpurun_out/j25_tests.log 2>&1 || { tail -40 gpurun_out/j25_tests.log; exit 1; }
tail -1 gpurun_out/j25_tests.log
one() {  # prob p tag env
  env $4 timeout -k 10 200 python -u bench.py --problem $1 --p $2 --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 30 --steps 50 --warmup 10 > gpurun_out/j25.json 2>gpurun_out/j25.err || { echo "$3 $1 p$2 FAILED"; tail -3 gpurun_out/j25.err; return 0; }
  python3 -c "
import json
r=json.loads(open('gpurun_out/j25.json').read().strip().splitlines()[-1])
print('$3', '$1 p$2', 'apply ms', round(r['ms_per_step'],4), 'frac', round(r['roofline_apply']['frac'],4), 'l2e ms', round(r['roofline']['avg_launch_ms'],4), 'cg ms/it', round(r['cg']['ms_per_iter'],4), 'final_nom', repr(r['cg']['final_nom']), flush=True)"
}
for prob in bp3 bp1 bp5; do
  for p in 4 5 6 7 8; do
    one $prob $p map MFEM_AMD_GATHER_CODE=0
    one $prob $p coded X=0
  done
done
