#!/usr/bin/env python3
"""Tuning aid: the two kernels of the apply ALTERNATING (as in BilinearForm::Mult), for a kernel trace
(rocprofv3 --kernel-trace): are the kernels as fast in that context as alone, and how long are the boundaries?
With --report <trace.csv> prints per (p) the average duration of each kernel and of the gaps between them."""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def report(path):
    import csv
    rows = [r for r in csv.DictReader(open(path)) if "k_pa_apply" in r["Kernel_Name"] or "k_scatter<0, false>" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # group consecutive (apply, scatter) pairs by the apply kernel's template arguments
    out = {}
    for i in range(len(rows) - 2):
        a, s, n = rows[i], rows[i + 1], rows[i + 2]
        if "k_pa_apply" in a["Kernel_Name"] and "k_scatter" in s["Kernel_Name"] and "k_pa_apply" in n["Kernel_Name"]:
            key = a["Kernel_Name"].split("(")[0].split("k_pa_apply")[-1]
            d = out.setdefault(key, [[], [], [], []])
            d[0].append(int(a["End_Timestamp"]) - int(a["Start_Timestamp"]))
            d[1].append(int(s["Start_Timestamp"]) - int(a["End_Timestamp"]))
            d[2].append(int(s["End_Timestamp"]) - int(s["Start_Timestamp"]))
            d[3].append(int(n["Start_Timestamp"]) - int(s["End_Timestamp"]))
    for k, d in out.items():
        m = [sorted(v[2:])[len(v[2:])//2]/1e3 if len(v) > 2 else float("nan") for v in d]
        print(f"{k}: n={len(d[0])} L->E {m[0]:7.1f} us | gap {m[1]:5.1f} | E->L {m[2]:6.1f} us | gap {m[3]:5.1f} | step {sum(m):7.1f} us")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--report":
        report(sys.argv[2]); sys.exit(0)
    import torch
    from mfem_amd import capi
    from mfem_amd.bp import BakeoffProblem
    torch.cuda.init()
    capi.call("mfem_amd_init", 0)
    dev = torch.device("cuda", 0)
    prob = sys.argv[1] if len(sys.argv) > 1 else "bp3"
    for p in [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "4,5,6,7,8").split(",")]:
        n = max(1, round(1.0e7**(1/3)/p))
        bp = BakeoffProblem(prob, p, (n, n, n), dev, device_tables=True)
        x = torch.rand(bp.ndofs, dtype=torch.float64, device=dev)
        y = torch.empty_like(x)
        for _ in range(25):
            bp.mult(x, y)
        torch.cuda.synchronize()
        del bp, x, y
        torch.cuda.empty_cache()
