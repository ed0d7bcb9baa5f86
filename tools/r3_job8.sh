#!/bin/bash
# round 3, job 8: E->L with greedy quarter grouping and as a kernel of its own: LDS tile size x register bound
set -o pipefail
O=gpurun_out/r3job8; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_direct_rows.py tests/test_gpu_subrows.py -x -q --timeout 300 --timeout-method thread > $O/parity.log 2>&1 || { tail -30 $O/parity.log; exit 1; }
tail -2 $O/parity.log
run() {  # lib tag problem p
  MFEM_AMD_LIB=$1 timeout -k 10 200 python -u bench.py --problem $3 --p $4 --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 0 --steps 50 --warmup 10 > $O/b.json 2>$O/b.err || { tail -5 $O/b.err; exit 1; }
  python3 -c "
import json
r=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
print('$2 $3 p$4 apply ms', round(r['ms_per_step'],4), 'frac', round(r['roofline_apply']['frac'],4), 'l2e', round(r['roofline']['avg_launch_ms'],4), 'e2l', round(r['roofline_apply']['e2l_avg_launch_ms'],4))
" | tee -a $O/summary.txt
}
run mfem_amd/libmfem_amd.so default bp3 4
for v in c3072_w4 c2048_w5 c2048_w6 c2048_w8 c2560_w6 c1536_w8; do
  run tools/variants/libmfem_amd_$v.so $v bp3 4
done
for v in c3072_w4 c2048_w6 c2048_w8 c2560_w6; do
  run tools/variants/libmfem_amd_$v.so $v bp1 4
  run tools/variants/libmfem_amd_$v.so $v bp3 2
  run tools/variants/libmfem_amd_$v.so $v bp3 6
done
