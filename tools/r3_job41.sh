#!/bin/bash
# PCG kernels A and B: 1 / 2 / 4 grid strides per trip with all loads of a trip requested first (same bits)
set -o pipefail
one() {  # lib
  MFEM_AMD_LIB=$PWD/mfem_amd/$1 timeout -k 10 400 python -u bench.py --no-cpu --no-cxx --no-orders --cg-iters 100 --steps 50 --warmup 10 > gpurun_out/j41.json 2>gpurun_out/j41.err || { tail -5 gpurun_out/j41.err; exit 1; }
  python3 -c "
import json
r=json.loads(open('gpurun_out/j41.json').read().strip().splitlines()[-1])
print('$1', 'bp3 p4 apply', round(r['roofline_apply']['frac'],4), 'cg ms/it', round(r['cg']['ms_per_iter'],4), 'cg frac', round(r['cg']['hbm_frac_per_gpu'],4), 'final_nom', repr(r['cg']['final_nom']))
print('   by_size cg_frac', ' '.join('%s:%.0e:%.3f' % (o['problem'], o['dofs'], o['cg_frac']) for o in r['by_size'] if o['dofs'] > 5e5), flush=True)"
}
one libprobe_pcgu1.so && one libprobe_pcgu2.so && one libmfem_amd.so && one libprobe_pcgu1.so && one libprobe_pcgu2.so && one libmfem_amd.so
