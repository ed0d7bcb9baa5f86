#!/bin/bash
# where the solver's streaming policy lands: per-kernel averages of a PCG-only run, policy off / on
O=$GRAFT_REPO_ROOT/gpurun_out/job68; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for nt in 0 1; do
  export MFEM_AMD_PCG_NT=$nt
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/nt$nt -o t -- python3 $GRAFT_REPO_ROOT/tools/pcg_only.py > $O/run$nt.log 2>&1 || { tail -20 $O/run$nt.log; exit 1; }
  f=$(find $O/nt$nt -name "*kernel_stats.csv" | head -1)
  echo "== MFEM_AMD_PCG_NT=$nt $(grep ms/iter $O/run$nt.log)" >> $O/summary.txt
  python3 - "$f" >> $O/summary.txt <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "mfa::" in r["Name"] and int(r["Calls"]) >= 100:
        print("  ", r["Name"][:60].ljust(60), r["Calls"].rjust(6), "avg", round(float(r["AverageNs"])/1e3,1), "us")
PY
  find $O/nt$nt -name "*.csv" ! -name "*kernel_stats.csv" -delete
done
cat $O/summary.txt
