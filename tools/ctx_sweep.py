#!/usr/bin/env python3
"""Tuning aid: the schedule / prefetch / contraction-level choices of the element kernel timed IN CONTEXT (the
alternating L->E, E->L kernels of the whole apply), one subprocess per setting (the overrides are read once per process).
Round 1's sweeps timed the element kernel in a loop of its own; the cache-policy work of round 2 showed that the two
can disagree.  usage: ctx_sweep.py [problem] [orders]"""
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
CODE = r'''
import sys, gc, torch
sys.path.insert(0, %r)
from mfem_amd import capi
from mfem_amd.bp import BakeoffProblem
torch.cuda.init(); capi.call("mfem_amd_init", 0)
dev = torch.device("cuda", 0)
prob, p = sys.argv[1], int(sys.argv[2])
n = max(1, round(1.0e7**(1/3)/p))
bp = BakeoffProblem(prob, p, (n, n, n), dev, device_tables=True)
x = torch.rand(bp.ndofs, dtype=torch.float64, device=dev); y = torch.empty_like(x)
for _ in range(3): bp.mult(x, y)
gc.collect(); torch.cuda.synchronize()
ts = []
for _ in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): bp.mult(x, y)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1)*1e3/20)
print("%%.1f" %% sorted(ts)[1])
''' % str(ROOT)


def run(prob, p, env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", CODE, prob, str(p)], env=e, capture_output=True, text=True, timeout=120)
    if r.returncode != 0:
        return "  n/a "
    return "%6s" % r.stdout.strip().splitlines()[-1]


if __name__ == "__main__":
    prob = sys.argv[1] if len(sys.argv) > 1 else "bp3"
    orders = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "4,5,6,7,8").split(",")]
    settings = [("table", {})]
    for mode in ("0", "2"):
        for pf in ("0", "2", "3"):
            for rb in ("1", "2"):
                settings.append((f"mode={mode} pf={pf} rb={rb}", {"MFEM_AMD_APPLY_MODE": mode, "MFEM_AMD_APPLY_PF": pf, "MFEM_AMD_APPLY_RB": rb}))
    print(f"{prob}: whole apply in us at ~1e7 dofs, p = {orders}", flush=True)
    for name, env in settings:
        print(f"{name:22s}", " ".join(run(prob, p, env) for p in orders), flush=True)
