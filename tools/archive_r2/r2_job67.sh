#!/bin/bash
# per-kernel averages under the C++ driver (reference numbering), BP3 p = 4 and 6
O=$GRAFT_REPO_ROOT/gpurun_out/job67; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for p in 4 6; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$p -o t -- $GRAFT_REPO_ROOT/examples/bin/bench_ceed -bp 3 -p $p -n $((216/p)) -mi 50 > $O/run$p.log 2>&1 || { tail -5 $O/run$p.log; exit 1; }
  f=$(find $O/p$p -name "*kernel_stats.csv" | head -1)
  echo "== p=$p" >> $O/summary.txt
  python3 - "$f" >> $O/summary.txt <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if int(r["Calls"]) >= 50: print(r["Name"][:60].ljust(60), r["Calls"].rjust(6), "avg", round(float(r["AverageNs"])/1e3,1), "us")
PY
  find $O/p$p -name "*.csv" ! -name "*kernel_stats.csv" -delete
done
cat $O/summary.txt
