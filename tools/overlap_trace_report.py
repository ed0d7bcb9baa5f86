#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace of ONE rank of a partitioned run: the kernels of one operator application in the steady
state, with start / end relative to the application's first kernel and the queue they ran on -- shows the exchange kernels
(pack / unpack / accumulate) beside the element kernel launches."""
import csv, sys
from pathlib import Path
f = next(Path(sys.argv[1]).rglob("*kernel_trace.csv"))
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.replace("void ", "").replace("mfa::", "").split("(")[0][:58]
apply_idx = [i for i, r in enumerate(rows) if "k_pa_apply<" in r["Kernel_Name"]]
# an application = two element-kernel launches (interior range, then interface range); take one from the middle of the timed block
per = int(sys.argv[2]) if len(sys.argv) > 2 else 2          # element-kernel launches per application
pairs = [apply_idx[k] for k in range(0, len(apply_idx) - per + 1, per)]
mid = pairs[len(pairs)//3]
nxt = pairs[len(pairs)//3 + 1]
lo = mid
while lo > 0 and "k_pa_apply<" not in rows[lo - 1]["Kernel_Name"] and int(rows[mid]["Start_Timestamp"]) - int(rows[lo - 1]["Start_Timestamp"]) < 300000:
    lo -= 1
t0 = int(rows[lo]["Start_Timestamp"])
print(f"trace: {f.name}; one application (kernels {lo}..{nxt - 1} of {len(rows)}), times in us from its first kernel")
print(f"{'kernel':60s} {'queue':>6s} {'start':>9s} {'end':>9s} {'dur':>8s}")
for r in rows[lo:nxt]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"{short(r['Kernel_Name']):60s} {r.get('Queue_Id', '?'):>6s} {s/1e3:9.1f} {e/1e3:9.1f} {(e - s)/1e3:8.1f}")
