#!/bin/bash
# Staged (LDS -> whole lines) stores of the element kernel against the direct per-row stores, p = 4; same bits expected (cg final_nom).
set -o pipefail
for lib in libmfem_amd.so libprobe_stage.so libmfem_amd.so libprobe_stage.so; do
  for prob in bp3 bp1 bp5; do
    MFEM_AMD_LIB=$PWD/mfem_amd/$lib timeout -k 10 200 python -u bench.py --problem $prob --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 30 --steps 50 --warmup 10 > gpurun_out/j14.json 2>gpurun_out/j14.err || { tail -5 gpurun_out/j14.err; exit 1; }
    python3 -c "
import json
r=json.loads(open('gpurun_out/j14.json').read().strip().splitlines()[-1])
print('$lib', '$prob', 'apply ms', round(r['ms_per_step'],4), 'frac', round(r['roofline_apply']['frac'],4), 'l2e ms', round(r['roofline']['avg_launch_ms'],4), 'e2l ms', round(r['roofline_apply']['e2l_avg_launch_ms'],4), 'cg ms/it', round(r['cg']['ms_per_iter'],4), 'final_nom', repr(r['cg']['final_nom']), flush=True)"
  done
done
