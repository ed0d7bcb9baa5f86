#!/bin/bash
# contraction level (RB 0 / 1 / 2) re-swept on the one-trip kernels, one batch per workgroup, whole columns: whole apply (us),
# ~1e7 dofs.  usage: r3_job40.sh <problem> <orders...>
prob=$1; shift
echo "$prob: whole apply (us), p = $*"
run() { # label env...
  label=$1; shift
  line=$(printf "%-14s" "$label")
  for p in $ORDERS; do
    v=$(env "$@" timeout -k 10 120 python -u bench.py --problem $prob --p $p --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 0 --steps 40 --warmup 10 2>/dev/null | python3 -c "
import json,sys
try:
    r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%6.1f' % (r['ms_per_step']*1e3))
except Exception: print('   n/a')")
    line="$line $v"
  done
  echo "$line"
}
ORDERS="$*"
run table X=0
for rb in 0 1 2; do run "rb=$rb" MFEM_AMD_APPLY_RB=$rb; done
run table X=0
