#!/bin/bash
# one-batch-per-workgroup kernels with a visible trip count (no preheader drain of the pa_data requests) against the
# library before it: whole apply by order, BP3 (A/B/A/B), BP1 and BP5 (A/B)
set -o pipefail
one() {  # lib problem
  MFEM_AMD_LIB=$PWD/mfem_amd/$1 timeout -k 10 300 python -u bench.py --problem $2 --no-cpu --no-cxx --no-sizes --cg-iters 30 --steps 50 --warmup 10 > gpurun_out/j37.json 2>gpurun_out/j37.err || { tail -5 gpurun_out/j37.err; exit 1; }
  python3 -c "
import json
r=json.loads(open('gpurun_out/j37.json').read().strip().splitlines()[-1])
print('$1', '$2', 'p4 apply ms', round(r['ms_per_step'],4), 'frac', round(r['roofline_apply']['frac'],4), 'l2e', round(r['roofline']['frac'],4), 'cg ms/it', round(r['cg']['ms_per_iter'],4), 'final_nom', repr(r['cg']['final_nom']))
print('   by_order apply', ' '.join('%.3f' % o['apply_frac'] for o in r['by_order']))
print('   by_order l2e  ', ' '.join('%.3f' % o['l2e_frac'] for o in r['by_order']), flush=True)"
}
one libprobe_base.so bp3 && one libprobe_onefull.so bp3 && one libprobe_base.so bp3 && one libprobe_onefull.so bp3 &&
one libprobe_base.so bp1 && one libprobe_onefull.so bp1 && one libprobe_base.so bp5 && one libprobe_onefull.so bp5
