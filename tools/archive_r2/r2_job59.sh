#!/bin/bash
set -o pipefail
O=gpurun_out/job59; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_gpu_direct_rows.py -x -q -m gpu --timeout 300 --timeout-method thread > $O/tests.log 2>&1 || { tail -25 $O/tests.log; exit 1; }
tail -2 $O/tests.log
timeout -k 10 500 python -u bench.py --no-cpu > $O/b.json 2> $O/b.err || { tail -20 $O/b.err; exit 1; }
python3 - $O/b.json <<'PY' | tee -a $O/summary.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", round(d["value"],3), "ms", round(d["ms_per_step"],4), "replay", round(d["graph_replay"]["ms_per_step"],4), "apply frac", round(d["roofline_apply"]["frac"],4), "cg", round(d["cg"]["ms_per_iter"],4), round(d["cg"]["hbm_frac_per_gpu"],4), "l2e", round(d["roofline"]["avg_launch_ms"],4), "e2l", round(d["roofline_apply"]["e2l_avg_launch_ms"],4))
print(" by_order us:", " ".join(str(round(r["ms_per_step"]*1e3,1)) for r in d["by_order"]))
print(" by_order frac:", " ".join(str(round(r["apply_frac"],3)) for r in d["by_order"]))
for r in d["by_size"]:
    print(" ", r["problem"], r["dofs"], "apply", round(r["ms_per_step"]*1e3,1), "us frac", round(r["apply_frac"],4), "cg_frac", round(r["cg_frac"],4))
PY
