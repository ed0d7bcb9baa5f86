#!/bin/bash
# final-tree evidence: rocprofv3 kernel stats of the bench command, then the PMC traffic passes for BP3/BP1/BP5 p=4
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/stats60; rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --no-cpu --no-orders --no-sizes --steps 20 --warmup 5 > $out/bench.log 2>&1 || { tail -5 $out/bench.log; exit 1; }
f=$(find $out -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/bench_p4_kernel_stats_v6.csv; head -8 $f | cut -c1-60,170-280
find $out -name "*.csv" ! -name "*kernel_stats.csv" -delete
bash tools/pmc_bench.sh p4c bp3_p4_n54 > /dev/null || exit 1
bash tools/pmc_bench.sh bp1p4c bp1_p4_n54 --problem bp1 > /dev/null || exit 1
bash tools/pmc_bench.sh bp5p4c bp5_p4_n54 --problem bp5 > /dev/null || exit 1
cat bench_traffic.json | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items(): print(k, round(v['l2e']['read_bytes']/1e6), round(v['l2e']['written_bytes']/1e6), round(v['e2l']['read_bytes']/1e6), round(v['e2l']['written_bytes']/1e6))"
