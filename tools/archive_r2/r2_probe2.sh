#!/bin/bash
set -e
out=gpurun_out/r2_probe2.txt
: > $out
timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -x -q -m gpu --timeout 120 --timeout-method thread >> $out 2>&1
echo "== parity suite with the block-cooperative E->L kernel" >> $out
MFEM_AMD_SCATTER_BLK=1 timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_bakeoff.py -x -q -m gpu --timeout 300 --timeout-method thread >> $out 2>&1
for blk in 0 1; do
  echo "== MFEM_AMD_SCATTER_BLK=$blk" >> $out
  for p in 1 2 3 4 6 8; do MFEM_AMD_SCATTER_BLK=$blk timeout -k 10 120 python -u tools/e2l_probe.py $p >> $out 2>&1; done
done
echo "== C++ API tests (generic CGSolver path)" >> $out
timeout -k 10 900 python -u -m pytest tests/test_gpu_cxx.py -x -q -m gpu --timeout 600 --timeout-method thread >> $out 2>&1
