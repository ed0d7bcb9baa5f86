#!/usr/bin/env python3
"""Average the rocprofv3 --pmc counter_collection CSVs per kernel (skipping warm-up dispatches)."""
import csv, sys, collections
from pathlib import Path
root = Path(sys.argv[1])
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in root.rglob("*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"].split("(")[0][-60:]
        acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = []
for k, d in acc.items():
    if "at::native" in k:
        continue
    out.append(f"== {k}")
    for c, v in sorted(d.items()):
        v = v[3:] if len(v) > 6 else v
        out.append(f"   {c:32s} n={len(v):3d} avg={sum(v)/len(v):.4g}")
txt = "\n".join(out)
print(txt)
(root / "summary.txt").write_text(txt + "\n")
