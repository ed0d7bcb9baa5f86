#!/bin/bash
# E->L over the shared rows only (interior rows direct): do the block / XCD choices of the full table still hold?
set -o pipefail
O=gpurun_out/job63; mkdir -p $O
for cfg in "base" "MFEM_AMD_SCATTER_XCD=1" "MFEM_AMD_SCATTER_BLK=0" "base" "MFEM_AMD_SCATTER_XCD=1"; do
  if [ "$cfg" = base ]; then envs=""; else envs="$cfg"; fi
  env $envs timeout -k 10 300 python -u bench.py --no-cpu --no-sizes --cg-iters 0 > $O/b.json 2> $O/b.err || { tail -5 $O/b.err; exit 1; }
  python3 - $O/b.json "$cfg" <<'PY' | tee -a $O/summary.txt
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2].ljust(26), "apply", round(d["ms_per_step"],4), "e2l", round(d["roofline_apply"]["e2l_avg_launch_ms"],4), "| by_order us:", " ".join(str(round(r["ms_per_step"]*1e3,1)) for r in d["by_order"]))
PY
done
