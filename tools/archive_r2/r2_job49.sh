#!/bin/bash
# PCG batch graph: parity tests, then A/B of the solve (eager one poll / eager polls of 10 / graph polls of 10).
set -o pipefail
mkdir -p gpurun_out/job49
O=gpurun_out/job49
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu --timeout 120 --timeout-method thread -k "pcg or constrained" > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
for mode in "0 100" "0 10" "1 10" "1 5" "1 20"; do
  set -- $mode
  echo "== PCG_GRAPH=$1 check=$2" | tee -a $O/ab.log
  MFEM_AMD_PCG_GRAPH=$1 MFEM_AMD_BENCH_CG_CHECK=$2 timeout -k 10 300 python -u bench.py --no-orders --no-cpu > $O/b_$1_$2.json 2> $O/b_$1_$2.err || { tail -20 $O/b_$1_$2.err; exit 1; }
  python3 - $O/b_$1_$2.json <<'PY' | tee -a $O/ab.log
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("apply ms", round(d["ms_per_step"],4), "replay", d["graph_replay"], "cg ms/iter", round(d["cg"]["ms_per_iter"],4), "frac", round(d["cg"]["hbm_frac_per_gpu"],4))
for r in d["by_size"]:
    print(r["problem"], r["dofs"], "apply", round(r["ms_per_step"]*1e3,1), "us cg_frac", round(r["cg_frac"],4), "cg Mdofs-it/s", round(r["cg_dofs_iter_per_s"]/1e6,1))
PY
done
