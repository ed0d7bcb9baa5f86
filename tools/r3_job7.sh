#!/bin/bash
# round 3, job 7: record the exact cg32 counts; C++ parallel bake-off, ex1p and the partitioned tests after the
# tolerance removal; BP5 traffic entry; BP1 with the XCD-aware element order
set -o pipefail
O=gpurun_out/r3job7; mkdir -p $O
MFEM_AMD_RECORD=1 timeout -k 10 300 python -u -m pytest tests/test_gpu_vector_ref.py -x -q --timeout 200 --timeout-method thread > $O/vector_ref_record.log 2>&1 || { tail -30 $O/vector_ref_record.log; exit 1; }
cp tests/golden/vec/cg32_device_counts.json $O/ && cat $O/cg32_device_counts.json | head -40
timeout -k 10 300 python -u -m pytest tests/test_gpu_vector_ref.py -x -q --timeout 200 --timeout-method thread > $O/vector_ref.log 2>&1 || { tail -30 $O/vector_ref.log; exit 1; }
tail -2 $O/vector_ref.log
timeout -k 10 900 python -u -m pytest tests/test_gpu_cxx.py tests/test_gpu_parallel.py -x -q --timeout 600 --timeout-method thread > $O/cxx_par.log 2>&1 || { tail -40 $O/cxx_par.log; exit 1; }
tail -2 $O/cxx_par.log
bash tools/pmc_bench.sh r3bp5p4 bp5_p4_n54 --problem bp5 > $O/pmc_bp5.log 2>&1 || { tail -20 $O/pmc_bp5.log; exit 1; }
tail -12 $O/pmc_bp5.log
for x in 0 1; do
  MFEM_AMD_APPLY_XCD=$x timeout -k 10 200 python -u bench.py --problem bp1 --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 0 --steps 50 --warmup 10 > $O/bp1_xcd$x.json 2>$O/bp1_xcd$x.err || { tail -5 $O/bp1_xcd$x.err; exit 1; }
  python3 -c "
import json
r=json.loads(open('$O/bp1_xcd$x.json').read().strip().splitlines()[-1])
print('bp1 xcd=$x apply ms', round(r['ms_per_step'],4), 'frac', round(r['roofline_apply']['frac'],4), 'l2e', round(r['roofline']['avg_launch_ms'],4), 'e2l', round(r['roofline_apply']['e2l_avg_launch_ms'],4))
" | tee -a $O/summary.txt
done
