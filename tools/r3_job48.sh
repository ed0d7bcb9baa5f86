#!/bin/bash
# mass kernel (7,8): 4 elements per workgroup (256 threads, 2048-byte runs per load) instead of 2 -- BP1 p = 6
one() { # lib label
  v=$(env MFEM_AMD_LIB=$PWD/mfem_amd/$1 timeout -k 10 120 python -u bench.py --problem bp1 --p 6 --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 20 --steps 40 --warmup 10 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%7.1f us  %.4f  kernel %.4f  final_nom %r' % (r['ms_per_step']*1e3, r['roofline_apply']['frac'], r['roofline']['frac'], r['cg']['final_nom']))")
  echo "bp1 p=6 $2: $v"
}
for rep in 1 2 3; do one libmfem_amd.so "nbz 2"; one libprobe_nbz8m_d7.so "nbz 4"; done
