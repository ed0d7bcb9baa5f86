#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/apply_trace; rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 tools/apply_trace.py bp3 3,4,5,6,7,8 > $out/run.log 2>&1 || { tail -5 $out/run.log; exit 1; }
f=$(find $out -name "*kernel_trace.csv" | head -1)
python3 tools/apply_trace.py --report $f | tee $out/report.txt
