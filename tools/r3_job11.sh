#!/bin/bash
# BP1 (mass) element kernel at p = 4, 6: schedule (MFEM_AMD_APPLY_MODE), streamed pa_data (PF), cache policy (NT) in context.
set -o pipefail
run() {  # tag, env...
  tag=$1; shift
  for p in 4 6; do
    env "$@" timeout -k 10 120 python3 bench.py --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 0 --steps 50 --warmup 10 --problem bp1 --p $p > gpurun_out/bp1sweep_tmp.json || { echo "$tag p=$p failed"; return 0; }
    python3 - "$tag" $p <<'PY'
import json, sys
d = json.loads(open('gpurun_out/bp1sweep_tmp.json').read().strip().splitlines()[-1])
print(sys.argv[1], 'bp1 p', sys.argv[2], 'apply ms', round(d['ms_per_step'], 4), 'frac', round(d['roofline_apply']['frac'], 4),
      'l2e', round(d['roofline']['avg_launch_ms'], 4), 'e2l', round(d['roofline_apply']['e2l_avg_launch_ms'], 4), flush=True)
PY
  done
}
run default X=0
run mode1 MFEM_AMD_APPLY_MODE=1
run mode2 MFEM_AMD_APPLY_MODE=2
run pf2 MFEM_AMD_APPLY_PF=2
run pf2_mode2 MFEM_AMD_APPLY_PF=2 MFEM_AMD_APPLY_MODE=2
run pf3_mode2 MFEM_AMD_APPLY_PF=3 MFEM_AMD_APPLY_MODE=2
run nt0 MFEM_AMD_APPLY_NT=0
run rb1 MFEM_AMD_APPLY_RB=1
run rb1_mode2 MFEM_AMD_APPLY_RB=1 MFEM_AMD_APPLY_MODE=2
run default2 X=0
