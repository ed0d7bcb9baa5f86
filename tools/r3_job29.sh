#!/bin/bash
# transpose-gather (block-cooperative form, apply path without the dot): workgroups of 64 / 128 / 512 threads (4 rows a thread, 12 entries a thread of LDS tile) against 256
set -o pipefail
for lib in libmfem_amd.so libprobe_tpb128.so libprobe_tpb64.so libprobe_tpb512.so libmfem_amd.so; do
for pp in "bp3 4" "bp3 2" "bp3 6" "bp1 4"; do set -- $pp
  MFEM_AMD_LIB=$PWD/mfem_amd/$lib timeout -k 10 200 python -u bench.py --problem $1 --p $2 --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 0 --steps 50 --warmup 10 > gpurun_out/j29.json 2>gpurun_out/j29.err || { tail -5 gpurun_out/j29.err; exit 1; }
  python3 -c "
import json
r=json.loads(open('gpurun_out/j29.json').read().strip().splitlines()[-1])
print('$lib', '$1 p$2', 'apply ms', round(r['ms_per_step'],4), 'frac', round(r['roofline_apply']['frac'],4), 'l2e ms', round(r['roofline']['avg_launch_ms'],4), 'e2l ms', round(r['roofline_apply']['e2l_avg_launch_ms'],4), flush=True)"
done
done
