#!/bin/bash
# per-kernel times of the bench with and without direct interior rows (interior-last numbering)
set -o pipefail
O=gpurun_out/job56; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for dr in 1 0; do
  export MFEM_AMD_DIRECT_ROWS=$dr
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/dr$dr -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-orders --no-sizes --no-cpu > $GRAFT_REPO_ROOT/$O/b$dr.json 2> $GRAFT_REPO_ROOT/$O/b$dr.err || { tail -20 $GRAFT_REPO_ROOT/$O/b$dr.err; exit 1; }
  f=$(find $GRAFT_REPO_ROOT/$O/dr$dr -name "*kernel_stats.csv" | head -1)
  echo "== direct=$dr" >> $GRAFT_REPO_ROOT/$O/summary.txt
  python3 - "$f" >> $GRAFT_REPO_ROOT/$O/summary.txt <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print(r["Name"][:70].ljust(70), r["Calls"].rjust(6), "avg", round(float(r["AverageNs"])/1e3,1), "us  total", round(float(r["TotalDurationNs"])/1e6,1), "ms")
PY
  find $GRAFT_REPO_ROOT/$O/dr$dr -name "*.csv" ! -name "*kernel_stats.csv" -delete
done
cat $GRAFT_REPO_ROOT/$O/summary.txt
