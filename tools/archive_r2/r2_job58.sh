#!/bin/bash
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/job58; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for dr in 1 0; do
  export MFEM_AMD_DIRECT_ROWS=$dr
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dr$dr -o t -- python3 $GRAFT_REPO_ROOT/tools/pcg_only.py > $O/run$dr.log 2>&1 || { tail -20 $O/run$dr.log; exit 1; }
  f=$(find $O/dr$dr -name "*kernel_stats.csv" | head -1)
  echo "== direct=$dr $(grep ms/iter $O/run$dr.log)" >> $O/summary.txt
  python3 - "$f" >> $O/summary.txt <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "mfa::" in r["Name"] and int(r["Calls"]) >= 100:
        print(r["Name"][:60].ljust(60), r["Calls"].rjust(6), "avg", round(float(r["AverageNs"])/1e3,1), "us")
PY
  find $O/dr$dr -name "*.csv" ! -name "*kernel_stats.csv" -delete
done
cat $O/summary.txt
