#!/bin/bash
# transpose-gather: LDS tile of the block-cooperative form, 3072 entries (shipped) / 4096 / 5120 -- p = 2 blocks hold ~3700 entries
# and go through in four parts today
one() { # lib prob p
  v=$(env MFEM_AMD_LIB=$PWD/mfem_amd/$1 timeout -k 10 120 python -u bench.py --problem $2 --p $3 --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 20 --steps 50 --warmup 10 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%7.1f us  %.4f  e2l %.1f us  cg %.4f ms/it  final_nom %r' % (r['ms_per_step']*1e3, r['roofline_apply']['frac'], r['roofline_apply']['e2l_avg_launch_ms']*1e3, r['cg']['ms_per_iter'], r['cg']['final_nom']))")
  echo "$2 p=$3 $1: $v"
}
for pp in "bp3 2" "bp3 3" "bp3 4" "bp1 4" "bp5 2" "bp3 6"; do set -- $pp
for lib in libmfem_amd.so libprobe_cap4096.so libprobe_cap5120.so libmfem_amd.so; do one $lib $1 $2; done
done
