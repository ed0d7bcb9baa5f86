#!/bin/bash
# (9,10) whole columns, one batch per workgroup (the new table entry): staged stores on (shipped build) / off (probe); then the
# schedule sweep at p = 5, 6, 7 on the one-trip kernels
one() { # lib label
  v=$(env MFEM_AMD_LIB=$PWD/mfem_amd/$1 MFEM_AMD_APPLY_MODE=0 MFEM_AMD_APPLY_PF=0 MFEM_AMD_APPLY_RB=2 timeout -k 10 120 python -u bench.py --problem $3 --p 8 --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 0 --steps 40 --warmup 10 2>/dev/null | python3 -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%6.1f' % (r['ms_per_step']*1e3))")
  echo "$3 p=8 mode 0 pf 0 rb 2 $2: $v us"
}
for rep in 1 2; do one libmfem_amd.so staged bp3; one libprobe_d9stage0.so direct bp3; one libmfem_amd.so staged bp1; one libprobe_d9stage0.so direct bp1; done
bash tools/r3_job33.sh bp3 5 6 7 && bash tools/r3_job33.sh bp5 5 6 7
