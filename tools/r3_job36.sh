#!/bin/bash
# kernel breakdown of a PCG iteration at 1e8 dofs (BP3 p = 4, n = 116): which of the vector kernels lags
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_pcg1e8 -- python3 bench.py --elems 116 --steps 10 --warmup 2 --cg-iters 30 --no-orders --no-sizes --no-cpu --no-cxx > gpurun_out/pcg1e8.json 2> gpurun_out/pcg1e8.err || { tail -5 gpurun_out/pcg1e8.err; exit 1; }
f=$(ls gpurun_out/prof_pcg1e8/*/*kernel_stats.csv | head -1)
cp $f gpurun_out/pcg1e8_kernel_stats.csv
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/pcg1e8_kernel_stats.csv')))
for r in rows[:14]:
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(6), ('%.1f'%(float(r['AverageNs'])/1e3)).rjust(10), r['Percentage'])
PY
