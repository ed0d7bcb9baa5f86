#!/bin/bash
# PCG batch graph, long solves: does the replay pay once the recording is amortised?
set -o pipefail
O=gpurun_out/job50; mkdir -p $O
for mode in "0 10 500" "1 10 500" "0 10 500" "1 10 500" "1 25 500" "1 10 2000" "0 10 2000"; do
  set -- $mode
  MFEM_AMD_PCG_GRAPH=$1 MFEM_AMD_BENCH_CG_CHECK=$2 timeout -k 10 300 python -u bench.py --no-orders --no-cpu --no-sizes --cg-iters $3 > $O/b.json 2> $O/b.err || { tail -20 $O/b.err; exit 1; }
  python3 - $O/b.json "$mode" <<'PY' | tee -a $O/ab.log
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "cg ms/iter", round(d["cg"]["ms_per_iter"],4), "frac", round(d["cg"]["hbm_frac_per_gpu"],4))
PY
done
