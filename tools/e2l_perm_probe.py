#!/usr/bin/env python3
"""Probe (tuning aid): E->L with the rows visited in mesh order instead of L-dof order.  MFEM numbers vertex, edge,
face and interior dofs in four separate sweeps over the mesh, so a 128-B line of the E-vector is needed at four
different times (measured 1.66x over-fetch, bench_traffic.json).  Here the rows are sorted by the lowest element that
touches them (a setup-time permutation, 4 B/row) and summed by the row-list kernel of
mfem_amd_restrict_scatter_subrows; y is bit-identical.  Prints the time against the shipped kernel."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def run(p, dofs):
    import torch
    from mfem_amd import capi
    from mfem_amd.cartesian import CartesianProblem
    dev = torch.device("cuda", 0)
    n = max(1, round(dofs**(1/3)/p))
    prob = CartesianProblem(p, n, n, n)
    D3, ne, nd = (p + 1)**3, prob.ne, prob.ndofs
    off, ind = prob.offsets.astype(np.int64), prob.indices.astype(np.int64)
    cnt = np.diff(off)
    first_el = ind[off[:-1]]//D3                       # entries of a row are in element order: the first is the lowest
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dt)).to(dev)
    ptr = lambda x: C.c_void_p(x.data_ptr())
    xE = torch.rand(ne*D3, dtype=torch.float64, device=dev)
    y0, y1 = torch.empty(nd, dtype=torch.float64, device=dev), torch.empty(nd, dtype=torch.float64, device=dev)
    d_off, d_ind = t(prob.offsets, np.int32), t(prob.indices, np.int32)

    def timed(fn, k=20):
        import gc
        gc.collect(); torch.cuda.synchronize()
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1)*1e3/k

    base = lambda: capi.call("mfem_amd_restrict_scatter", nd, ptr(d_off), ptr(d_ind), ptr(xE), ptr(y0), 0, 0, None)
    tb = timed(base)
    out = [f"p={p} n={n} ndofs={nd}: shipped {tb:6.1f} us"]
    for name, perm in (("identity", np.arange(nd)), ("by lowest element", np.argsort(first_el, kind="stable")),
                       ("by lowest element, 8-element groups", np.argsort(first_el//8, kind="stable"))):
        pc = cnt[perm]
        poff = np.zeros(nd + 1, np.int64); np.cumsum(pc, out=poff[1:])
        # entries of permuted row k = entries of row perm[k], same order
        src = np.repeat(off[:-1][perm] - poff[:-1], pc) + np.arange(poff[-1])
        pind = ind[src]
        d_rows, d_poff, d_pind = t(perm, np.int32), t(poff, np.int32), t(pind, np.int32)
        fn = lambda: capi.call("mfem_amd_restrict_scatter_subrows", nd, ptr(d_rows), ptr(d_poff), ptr(d_pind), ptr(xE), ptr(y1), None)
        y1.fill_(float("nan"))
        tp = timed(fn)
        same = bool(torch.equal(y0, y1))
        out.append(f"rows {name}: {tp:6.1f} us (bit-identical {same})")
    print(" | ".join(out), flush=True)


if __name__ == "__main__":
    import torch
    from mfem_amd import capi
    torch.cuda.init()
    capi.call("mfem_amd_init", 0)
    for p in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "2,4,6,8").split(",")]:
        run(p, 1e7)
