#!/bin/bash
# Final-state measurements of the round: PMC traffic of the two apply kernels (BP3 / BP1 / BP5 at p = 4; BP3 p = 5, 8 with the
# in-place / whole-column kernels), rocprofv3 kernel stats of the default bench command
set -o pipefail
bash tools/pmc_bench.sh r3t_p4 bp3_p4_n54 > gpurun_out/pmc_bench_r3t_p4.txt || exit 1
bash tools/pmc_bench.sh r3t_bp1p4 bp1_p4_n54 --problem bp1 > gpurun_out/pmc_bench_r3t_bp1p4.txt || exit 1
bash tools/pmc_bench.sh r3t_bp5p4 bp5_p4_n54 --problem bp5 > gpurun_out/pmc_bench_r3t_bp5p4.txt || exit 1
bash tools/pmc_bench.sh r3t_p5 bp3_p5_n43 --p 5 > gpurun_out/pmc_bench_r3t_p5.txt || exit 1
bash tools/pmc_bench.sh r3t_p8 bp3_p8_n27 --p 8 > gpurun_out/pmc_bench_r3t_p8.txt || exit 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_v4 -- python3 bench.py --no-cpu --no-cxx --no-orders --no-sizes > gpurun_out/stats_v4.log 2>&1 || { tail -5 gpurun_out/stats_v4.log; exit 1; }
find gpurun_out/stats_v4 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/bench_p4_kernel_stats_v4.csv
find gpurun_out/stats_v4 -name "*.csv" ! -name "*kernel_stats.csv" -delete
head -8 gpurun_out/bench_p4_kernel_stats_v4.csv | cut -c1-100
cp bench_traffic.json gpurun_out/bench_traffic.json
tail -4 gpurun_out/pmc_bench_r3t_p4.txt
