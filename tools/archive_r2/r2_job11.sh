#!/bin/bash
# occupancy probe: how does the fused L->E kernel react to extra LDS per workgroup (the budget of a block-local E-vector)?
export MFEM_AMD_LIB=$PWD/mfem_amd/libmfem_amd_probe.so
out=gpurun_out/lds_pad_probe.txt; : > $out
for pad in 0 8192 16384 24576 32768 49152 65536; do
  echo "== pad $pad" >> $out
  MFEM_AMD_APPLY_LDS_PAD=$pad timeout -k 10 120 python3 tools/kbench.py --problem bp3,bp1,bp5 --p 4 --dofs 1e7 --which l2e >> $out 2>&1 || exit 1
done
cat $out
