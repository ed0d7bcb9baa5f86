#!/bin/bash
bash tools/pmc_bench.sh p4b bp3_p4_n54 || exit 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/stats30; rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --no-cpu --no-orders --no-sizes --steps 20 --warmup 5 > $out/bench.log 2>&1 || { tail -5 $out/bench.log; exit 1; }
f=$(find $out -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/bench_p4_kernel_stats_v4.csv; head -8 $f | cut -c1-60,150-260
