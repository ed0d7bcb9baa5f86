#!/bin/bash
# Pipelined transpose-gather (k_scatter_pipe) at two register budgets against the plain form (MFEM_AMD_SCATTER_PIPE=0): parity, then times.
set -o pipefail
for lib in libprobe_pipe3.so libprobe_pipe4.so; do
MFEM_AMD_LIB=$PWD/mfem_amd/$lib timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_direct_rows.py tests/test_gpu_bakeoff.py -x -q --timeout 300 --timeout-method thread > gpurun_out/j20_tests.log 2>&1 || { tail -30 gpurun_out/j20_tests.log; exit 1; }
echo $lib; tail -1 gpurun_out/j20_tests.log
done
one() {  # lib prob p tag env...
  lib=$1; prob=$2; p=$3; tag=$4; shift 4
  env "$@" MFEM_AMD_LIB=$PWD/mfem_amd/$lib timeout -k 10 200 python -u bench.py --problem $prob --p $p --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 30 --steps 50 --warmup 10 > gpurun_out/j20.json 2>gpurun_out/j20.err || { echo "$tag $prob p$p FAILED"; tail -3 gpurun_out/j20.err; return 0; }
  python3 -c "
import json
r=json.loads(open('gpurun_out/j20.json').read().strip().splitlines()[-1])
print('$tag', '$prob p$p', 'apply ms', round(r['ms_per_step'],4), 'frac', round(r['roofline_apply']['frac'],4), 'l2e ms', round(r['roofline']['avg_launch_ms'],4), 'e2l ms', round(r['roofline_apply']['e2l_avg_launch_ms'],4), 'cg ms/it', round(r['cg']['ms_per_iter'],4), 'final_nom', repr(r['cg']['final_nom']), flush=True)"
}
for p in 4 2 3 6 8; do
  one libprobe_pipe3.so bp3 $p plain MFEM_AMD_SCATTER_PIPE=0
  one libprobe_pipe3.so bp3 $p pipe_w3 X=0
  one libprobe_pipe4.so bp3 $p pipe_w4 X=0
done
one libprobe_pipe3.so bp1 4 plain MFEM_AMD_SCATTER_PIPE=0
one libprobe_pipe3.so bp1 4 pipe_w3 X=0
one libprobe_pipe4.so bp1 4 pipe_w4 X=0
