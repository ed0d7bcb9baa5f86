#!/bin/bash
# p = 7, 8: depth of the streamed pa_data ring, PF = 2 .. 6 (D = 9: one wave per SIMD with the whole register file; D = 8: two waves or one)
set -o pipefail
one() { # lib prob p pf
  MFEM_AMD_LIB=$PWD/mfem_amd/$1 MFEM_AMD_APPLY_PF=$4 timeout -k 10 200 python -u bench.py --problem $2 --p $3 --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 30 --steps 50 --warmup 10 > gpurun_out/j32.json 2>gpurun_out/j32.err || { echo "$1 PF=$4 $2 p$3 failed"; tail -2 gpurun_out/j32.err; return 0; }
  python3 -c "
import json
r=json.loads(open('gpurun_out/j32.json').read().strip().splitlines()[-1])
print('$1 PF=$4', '$2 p$3', 'apply ms', round(r['ms_per_step'],4), 'frac', round(r['roofline_apply']['frac'],4), 'l2e ms', round(r['roofline']['avg_launch_ms'],4), 'cg ms/it', round(r['cg']['ms_per_iter'],4), 'final_nom', repr(r['cg']['final_nom']), flush=True)"
}
for pf in 2 3 4 5 6; do one libprobe_deep9.so bp3 8 $pf; done
for pf in 2 3 4 5 6; do one libprobe_deep9.so bp5 8 $pf; done
for pf in 2 3 4 5 6; do one libprobe_deep8w2.so bp3 7 $pf; done
for pf in 4 5 6; do one libprobe_deep8w1.so bp3 7 $pf; done
for pf in 2 3 4 5; do one libprobe_deep8w2.so bp5 7 $pf; done
for pf in 4 5 6; do one libprobe_deep8w1.so bp5 7 $pf; done
