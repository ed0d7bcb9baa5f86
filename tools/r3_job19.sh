#!/bin/bash
# Final-state measurements: PMC traffic of the two apply kernels (BP3 / BP1 / BP5, p = 4), rocprofv3 kernel stats of the default bench command, the bench line.
set -o pipefail
bash tools/pmc_bench.sh r3s_p4 bp3_p4_n54 || exit 1
bash tools/pmc_bench.sh r3s_bp1p4 bp1_p4_n54 --problem bp1 || exit 1
bash tools/pmc_bench.sh r3s_bp5p4 bp5_p4_n54 --problem bp5 || exit 1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/stats_v3 -- python3 bench.py --no-cpu --no-cxx --no-orders --no-sizes > gpurun_out/stats_v3.log 2>&1 || { tail -5 gpurun_out/stats_v3.log; exit 1; }
find gpurun_out/stats_v3 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/bench_p4_kernel_stats_v3.csv
head -8 gpurun_out/bench_p4_kernel_stats_v3.csv
cp bench_traffic.json gpurun_out/bench_traffic.json
timeout -k 10 900 python3 bench.py > gpurun_out/bench_p4_v3.log 2>&1 || { tail -5 gpurun_out/bench_p4_v3.log; exit 1; }
grep -a "^{" gpurun_out/bench_p4_v3.log | tail -1 > gpurun_out/bench_p4_v3.json
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/bench_p4_v3.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['roofline_apply']['frac'])
print(d['cg'])
print(' '.join('%.3f' % r['apply_frac'] for r in d['by_order']))
for r in d['by_size']: print(r['problem'], r['dofs'], round(r['apply_frac'],3), round(r.get('cg_frac',0),3))
print(d['cxx_drivers']); print(d['cpu_baseline'])
PY
