#!/bin/bash
# round 3, job 9: element kernel, runs of G consecutive batches per XCD (MFEM_AMD_APPLY_XCD=G)
set -o pipefail
O=gpurun_out/r3job9; mkdir -p $O
run() {  # G problem p
  MFEM_AMD_APPLY_XCD=$1 timeout -k 10 200 python -u bench.py --problem $2 --p $3 --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 0 --steps 50 --warmup 10 > $O/b.json 2>$O/b.err || { tail -5 $O/b.err; exit 1; }
  python3 -c "
import json
r=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
print('G=$1 $2 p$3 apply ms', round(r['ms_per_step'],4), 'frac', round(r['roofline_apply']['frac'],4), 'l2e', round(r['roofline']['avg_launch_ms'],4), 'e2l', round(r['roofline_apply']['e2l_avg_launch_ms'],4))
" | tee -a $O/summary.txt
}
for g in 0 4 8 16 32 64 0; do run $g bp3 4; done
for g in 0 4 8 16 32 0; do run $g bp1 4; done
for g in 0 8 32; do run $g bp3 2; run $g bp3 6; run $g bp3 8; done
