#!/bin/bash
# The dispatch table of the element kernel re-swept in context on the final tree (staged stores, packed E-vector, reference numbering):
# schedule x ring depth x contraction level through bench.py, whole apply in us at ~1e7 dofs.  usage: r3_job33.sh <problem> <orders...>
prob=$1; shift
echo "$prob: whole apply (us), p = $*"
run() { # label env...
  label=$1; shift
  line=$(printf "%-22s" "$label")
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
for mode in 0 2; do for pf in 0 2 3; do for rb in 1 2; do
  run "mode=$mode pf=$pf rb=$rb" MFEM_AMD_APPLY_MODE=$mode MFEM_AMD_APPLY_PF=$pf MFEM_AMD_APPLY_RB=$rb
done; done; done
run table X=0
