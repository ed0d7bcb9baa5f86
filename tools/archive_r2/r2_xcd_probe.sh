#!/bin/bash
# r2 probe: XCD-aware block -> work mapping of the element kernel (one batch per workgroup) and of the E->L kernel
set -e
out=gpurun_out/r2_xcd_probe.txt
: > $out
for sx in 0 1; do for ax in 0 1; do
  echo "== MFEM_AMD_SCATTER_XCD=$sx MFEM_AMD_APPLY_XCD=$ax" >> $out
  MFEM_AMD_SCATTER_XCD=$sx MFEM_AMD_APPLY_XCD=$ax timeout -k 10 300 python -u tools/kbench.py --problem bp3,bp1 --p 2,4,6 >> $out 2>&1
done; done
echo "== e2l_probe (scatter remap on)" >> $out
for p in 2 4 6; do timeout -k 10 120 python -u tools/e2l_probe.py $p >> $out 2>&1; done
echo "== e2l_probe (scatter remap off)" >> $out
for p in 2 4 6; do MFEM_AMD_SCATTER_XCD=0 timeout -k 10 120 python -u tools/e2l_probe.py $p >> $out 2>&1; done
