#!/bin/bash
# p = 3 and p = 7 (where the 16-byte staged stores lost): the 8-byte copy-out against the direct stores of the shipped library
set -o pipefail
for rep in 1 2; do
for cfg in "libmfem_amd.so 3" "libprobe_stage8_d4.so 3" "libmfem_amd.so 7" "libprobe_stage8_d8.so 7"; do set -- $cfg
for prob in bp3 bp5; do
  MFEM_AMD_LIB=$PWD/mfem_amd/$1 timeout -k 10 200 python -u bench.py --problem $prob --p $2 --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 0 --steps 50 --warmup 10 > gpurun_out/j30.json 2>gpurun_out/j30.err || { tail -5 gpurun_out/j30.err; exit 1; }
  python3 -c "
import json
r=json.loads(open('gpurun_out/j30.json').read().strip().splitlines()[-1])
print('$1', '$prob p$2', 'apply ms', round(r['ms_per_step'],4), 'frac', round(r['roofline_apply']['frac'],4), 'l2e ms', round(r['roofline']['avg_launch_ms'],4), flush=True)"
done
done
done
