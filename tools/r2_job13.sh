#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/stats13; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --no-cpu --no-orders --no-sizes --steps 20 --warmup 5 > $out/bench.log 2>&1 || { tail -5 $out/bench.log; exit 1; }
tail -1 $out/bench.log
f=$(find $out -name "*kernel_stats.csv" | head -1); cat $f
