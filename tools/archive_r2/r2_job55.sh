#!/bin/bash
set -o pipefail
O=gpurun_out/job55; mkdir -p $O
for num in lexicographic interior-last; do
  for dr in 1 0; do
    [ $num = lexicographic ] && [ $dr = 0 ] && continue
    echo "== numbering=$num direct=$dr" | tee -a $O/summary.txt
    MFEM_AMD_BENCH_NUMBERING=$num MFEM_AMD_DIRECT_ROWS=$dr timeout -k 10 500 python -u bench.py --no-cpu > $O/b_${num}_$dr.json 2> $O/b.err || { tail -20 $O/b.err; exit 1; }
    python3 - $O/b_${num}_$dr.json <<'PY' | tee -a $O/summary.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value", round(d["value"],3), "ms", round(d["ms_per_step"],4), "replay", round(d["graph_replay"]["ms_per_step"],4), "apply frac", round(d["roofline_apply"]["frac"],4), "cg", round(d["cg"]["ms_per_iter"],4), round(d["cg"]["hbm_frac_per_gpu"],4), "direct", d["config"]["element_interior_rows_written_by_element_kernel"])
print(" by_order us:", " ".join(str(round(r["ms_per_step"]*1e3,1)) for r in d["by_order"]))
print(" by_order frac:", " ".join(str(round(r["apply_frac"],3)) for r in d["by_order"]))
for r in d["by_size"]:
    print(" ", r["problem"], r["dofs"], "apply", round(r["ms_per_step"]*1e3,1), "us frac", round(r["apply_frac"],4), "cg_frac", round(r["cg_frac"],4))
PY
  done
done
