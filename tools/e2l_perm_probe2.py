#!/usr/bin/env python3
"""As e2l_perm_probe.py, on the REFERENCE'S numbering (MFEM's dof numbering + space-filling-curve element order, the tables
bench.py runs on): transpose-gather with the rows visited in the order of the lowest element that touches them (row-list
kernel of mfem_amd_restrict_scatter_subrows; y bit-identical) against the shipped kernel.  usage: e2l_perm_probe2.py <problem> <p>..."""
import ctypes as C
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench


def run(problem, p):
    import torch
    from mfem_amd import capi
    from mfem_amd.bp import BakeoffProblem
    dev = torch.device("cuda", 0)
    n = max(1, round((1.0e7)**(1/3)/p))
    bp = BakeoffProblem(problem, p, (n, n, n), dev, device_tables=True, numbering=bench.numbering_for((p*n + 1)**3, p))
    D3, ne, nd = bp.D**3, bp.ne, bp.ndofs
    off, ind = bp.off.cpu().numpy().astype(np.int64), bp.ind.cpu().numpy().astype(np.int64)
    cnt = np.diff(off)
    idx = np.where(ind >= 0, ind, -1 - ind)
    first_el = np.where(cnt > 0, idx[np.minimum(off[:-1], len(idx) - 1)]//D3, ne)
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dt)).to(dev)
    ptr = lambda x: C.c_void_p(x.data_ptr())
    xE = torch.rand(ne*D3, dtype=torch.float64, device=dev)
    y0, y1 = torch.empty(nd, dtype=torch.float64, device=dev), torch.empty(nd, dtype=torch.float64, device=dev)

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

    base = lambda: capi.call("mfem_amd_restrict_scatter", nd, ptr(bp.off), ptr(bp.ind), ptr(xE), ptr(y0), 0, 0, None)
    tb = timed(base)
    out = [f"{problem} p={p} n={n} ndofs={nd} entries/row={len(ind)/nd:.2f}: shipped {tb:6.1f} us"]
    for name, perm in (("identity", np.arange(nd)), ("by lowest element", np.argsort(first_el, kind="stable")),
                       ("by lowest element, 8-element groups", np.argsort(first_el//8, kind="stable")),
                       ("by lowest element, 64-element groups", np.argsort(first_el//64, kind="stable"))):
        pc = cnt[perm]
        poff = np.zeros(nd + 1, np.int64); np.cumsum(pc, out=poff[1:])
        src = np.repeat(off[:-1][perm] - poff[:-1], pc) + np.arange(poff[-1])
        pind = ind[src]
        d_rows, d_poff, d_pind = t(perm, np.int32), t(poff, np.int32), t(pind, np.int32)
        fn = lambda: capi.call("mfem_amd_restrict_scatter_subrows", nd, ptr(d_rows), ptr(d_poff), ptr(d_pind), ptr(xE), ptr(y1), None)
        y1.fill_(float("nan"))
        tp = timed(fn)
        same = bool(torch.equal(y0, y1))
        out.append(f"   rows {name:38s} {tp:6.1f} us  bit-identical {same}")
    print("\n".join(out), flush=True)


for a in sys.argv[2:]:
    run(sys.argv[1], int(a))
