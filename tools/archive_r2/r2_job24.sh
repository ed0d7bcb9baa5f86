#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for nt in 0 1; do
  out=gpurun_out/apply_trace_nt$nt; rm -rf $out; mkdir -p $out
  MFEM_AMD_SCATTER_NT=$nt timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 tools/apply_trace.py bp3 4,5,6,7,8 > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
  f=$(find $out -name "*kernel_trace.csv" | head -1)
  echo "== MFEM_AMD_SCATTER_NT=$nt"; python3 tools/apply_trace.py --report $f | tee $out/report.txt
done
