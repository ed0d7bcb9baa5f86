#!/bin/bash
# the C++ drivers (MFEM numbering / element order) at ~1e7 dofs, p = 1..8, BP1 / BP3 / BP5 on the last tree of round 3 (as tools/archive_r2/r2_job64.sh)
set -o pipefail
O=gpurun_out/job50; mkdir -p $O
for bp in 3 1 5; do
  for p in 1 2 3 4 5 6 7 8; do
    n=$((216/p))
    timeout -k 10 300 examples/bin/bench_ceed -bp $bp -p $p -n $n -mi 50 >> $O/sweep_bp${bp}.jsonl 2>> $O/err.log || { tail -5 $O/err.log; exit 1; }
  done
done
python3 - $O <<'PY' | tee $O/summary.txt
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/sweep_bp*.jsonl")):
    rows=[json.loads(l) for l in open(f) if l.startswith("{")]
    k=[r for r in rows if r["phase"]=="kernel"]; q=[r for r in rows if r["phase"]=="problem"]
    print(f.split("/")[-1], "apply frac:", " ".join(f'{r["hbm_frac"]:.3f}' for r in k))
    print("   ms/apply:", " ".join(f'{r["ms_per_apply"]:.4f}' for r in k))
    print("   PCG frac:", " ".join(f'{r["hbm_frac"]:.3f}' for r in q), " iterations:", " ".join(str(r["iterations"]) for r in q))
PY
