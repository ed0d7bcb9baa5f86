#!/bin/bash
# in-place transposed y contraction (no LDS region V; diffusion, RB = 1, one batch per workgroup) forced on everywhere
# (libprobe_inpl.so, -DMFA_INPLACE=1) against the shipped library: parity tests on the probe, then whole apply by order
set -o pipefail
MFEM_AMD_LIB=$PWD/mfem_amd/libprobe_inpl.so timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_direct_rows.py tests/test_gpu_bakeoff.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/j38_tests.log 2>&1 || { tail -30 gpurun_out/j38_tests.log; exit 1; }
tail -2 gpurun_out/j38_tests.log
one() {  # lib problem env...
  lib=$1; prob=$2; shift; shift
  env MFEM_AMD_LIB=$PWD/mfem_amd/$lib "$@" timeout -k 10 300 python -u bench.py --problem $prob --no-cpu --no-cxx --no-sizes --cg-iters 30 --steps 50 --warmup 10 > gpurun_out/j38.json 2>gpurun_out/j38.err || { tail -5 gpurun_out/j38.err; exit 1; }
  python3 -c "
import json
r=json.loads(open('gpurun_out/j38.json').read().strip().splitlines()[-1])
print('$lib', '$prob', '$*', 'p4 apply ms', round(r['ms_per_step'],4), 'frac', round(r['roofline_apply']['frac'],4), 'cg ms/it', round(r['cg']['ms_per_iter'],4), 'final_nom', repr(r['cg']['final_nom']))
print('   by_order apply', ' '.join('%.3f' % o['apply_frac'] for o in r['by_order']))
print('   by_order l2e  ', ' '.join('%.3f' % o['l2e_frac'] for o in r['by_order']), flush=True)"
}
one libmfem_amd.so bp3 X=0 && one libprobe_inpl.so bp3 X=0 && one libmfem_amd.so bp3 X=0 && one libprobe_inpl.so bp3 X=0 &&
one libmfem_amd.so bp5 X=0 && one libprobe_inpl.so bp5 X=0 && one libmfem_amd.so bp5 MFEM_AMD_APPLY_RB=1 && one libprobe_inpl.so bp5 MFEM_AMD_APPLY_RB=1
