#!/bin/bash
set -o pipefail
O=gpurun_out/job53; mkdir -p $O
timeout -k 10 500 python -u bench.py --no-cpu > $O/b.json 2> $O/b.err || { tail -20 $O/b.err; exit 1; }
python3 - $O/b.json <<'PY' | tee $O/summary.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "replay", d["graph_replay"])
print("roofline", d["roofline"]["frac"], "apply", d["roofline_apply"]["frac"], "cg", d["cg"]["ms_per_iter"], d["cg"]["hbm_frac_per_gpu"])
for r in d["by_order"]:
    print("p", r["p"], round(r["ms_per_step"]*1e3,1), "us apply_frac", round(r["apply_frac"],4), "l2e_frac", round(r["l2e_frac"],4))
for r in d["by_size"]:
    print(r["problem"], r["dofs"], "apply", round(r["ms_per_step"]*1e3,1), "us frac", round(r["apply_frac"],4), "cg_frac", round(r["cg_frac"],4), r["launch"])
PY
