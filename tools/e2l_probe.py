#!/usr/bin/env python3
"""Tuning aid: how much of the E->L kernel's time is the scattered E-vector gather?  Times
mfem_amd_restrict_scatter on the real tables and on the same row structure with the index list
replaced by 0,1,2,... (perfectly coalesced reads of the same number of bytes)."""
import ctypes as C, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mfem_amd import capi
from mfem_amd.cartesian import CartesianProblem
torch.cuda.init(); capi.call("mfem_amd_init", 0)
dev = torch.device("cuda", 0)
p = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = round(1e7**(1/3)/p)
prob = CartesianProblem(p, n, n, n)
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
ptr = lambda x: C.c_void_p(x.data_ptr())
off, ind = t(prob.offsets), t(prob.indices)
iota = torch.arange(ind.numel(), dtype=torch.int32, device=dev)
perm = torch.randperm(ind.numel(), device=dev).to(torch.int32)
yE = torch.rand(ind.numel(), dtype=torch.float64, device=dev); y = torch.empty(prob.ndofs, dtype=torch.float64, device=dev)
def timed(idx):
    f = lambda: capi.call("mfem_amd_restrict_scatter", prob.ndofs, ptr(off), ptr(idx), ptr(yE), ptr(y), 0, 0, None)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/20*1e3
print(f"p={p} ndofs={prob.ndofs}: real tables {timed(ind):.1f} us | sequential indices {timed(iota):.1f} us | random permutation {timed(perm):.1f} us")
