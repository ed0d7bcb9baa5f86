#!/bin/bash
# D = 5 vector kernels: 3 waves/SIMD with 60-96 B of scratch per lane (A) against 2 waves without (B)
out=gpurun_out/d5_spill_probe.txt; : > $out
for v in A B; do
  echo "== variant $v" >> $out
  MFEM_AMD_LIB=$PWD/mfem_amd/libmfem_amd_probe$v.so timeout -k 10 200 python3 tools/kbench.py --problem bp4,bp6,bp2 --p 4 --vdim 3 --dofs 1e7 >> $out 2>&1 || exit 1
  MFEM_AMD_LIB=$PWD/mfem_amd/libmfem_amd_probe$v.so timeout -k 10 200 python3 tools/kbench.py --problem bp3 --p 4 --layout 0,1 --dofs 1e7 >> $out 2>&1 || exit 1
done
grep -v amdgpu.ids $out
