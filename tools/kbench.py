#!/usr/bin/env python3
"""Kernel microbenchmark (tuning aid): times the fused L->E kernel and the E->L kernel for
one (problem, p, n) with HIP events; one line per case.  Used under rocprofv3 --pmc too."""
import argparse
import ctypes as C
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402
from mfem_amd import tables  # noqa: E402


def run(problem, p, n, steps, which, layout=1, vdim=0, byvdim=1):
    import torch
    from mfem_amd import capi
    from mfem_amd.cartesian import CartesianProblem
    dev = torch.device("cuda", 0)
    D = p + 1
    kind = 0 if problem == "bp1" else 1
    Q = p + 1 if problem == "bp5" else p + 2
    prob = CartesianProblem(p, n, n, n)
    ne, ndofs = prob.ne, prob.ndofs
    kind, D, Q, B, G, Bn, Gn, W = tables.bakeoff_tables(problem, p)
    t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a, dt)).to(dev)
    ptr = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None
    gm, off, ind = t(prob.gather_map), t(prob.offsets), t(prob.indices)
    nodesE, Wd = t(prob.nodes_E), t(W)
    LAYOUT = layout
    pa = torch.zeros(capi.lib.mfem_amd_pa_data_size(kind, LAYOUT, Q, ne), dtype=torch.float64, device=dev)
    one = torch.ones(1, dtype=torch.float64, device=dev)
    capi.call("mfem_amd_pa_setup_from_nodes_3d", kind, 2, Q, 1, ne, Bn, Gn, ptr(Wd), ptr(nodesE), ptr(one), 1,
              layout, ptr(pa), None, None)
    x = torch.rand(ndofs, dtype=torch.float64, device=dev)
    y = torch.empty_like(x)
    yE = torch.empty(ne*D**3, dtype=torch.float64, device=dev)

    if vdim:
        # vector space (BP2/BP4/BP6): vdim components, one shared pa_data
        x = torch.rand(ndofs*vdim, dtype=torch.float64, device=dev)
        y = torch.empty_like(x)
        yE = torch.empty(ne*D**3*vdim, dtype=torch.float64, device=dev)

    def l2e():
        if vdim:
            capi.call("mfem_amd_pa_apply_L2E_vdim", kind, LAYOUT, D, Q, ne, vdim, byvdim, ndofs, B, G, ptr(pa), 0, ptr(gm), None,
                      ptr(x), ptr(yE), None)
        else:
            capi.call("mfem_amd_pa_apply_L2E", kind, LAYOUT, D, Q, ne, B, G, ptr(pa), ptr(gm), ptr(x), ptr(yE), None)

    def e2l():
        if vdim:
            capi.call("mfem_amd_restrict_scatter_vdim", ndofs, D**3, vdim, byvdim, ptr(off), ptr(ind), ptr(yE), ptr(y), 0, 0, None)
        else:
            capi.call("mfem_amd_restrict_scatter", ndofs, ptr(off), ptr(ind), ptr(yE), ptr(y), 0, 0, None)

    def timed(fn, k):
        import gc
        if os.environ.get("KBENCH_GC", "1") != "0":
            gc.collect()                  # tensors of the previous case that sit in reference cycles are freed NOW,
            torch.cuda.synchronize()      # not by a collection in the middle of the timed launches
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)*1e-3/k

    if os.environ.get("KBENCH_PTRS"):
        print("   ptrs pa=%x x=%x yE=%x gm=%x" % (pa.data_ptr(), x.data_ptr(), yE.data_ptr(), gm.data_ptr()), flush=True)
    b1, b2 = bench.algorithmic_bytes(kind, D, Q, ne, ndofs)
    if vdim:
        # pa_data, gather_map, indices and offsets once; x read and y written per component
        b1 += (vdim - 1)*8*ndofs
        b2 += (vdim - 1)*8*ndofs
    t1 = timed(l2e, steps) if which in ("both", "l2e") else float("nan")
    t2 = timed(e2l, steps) if which in ("both", "e2l") else float("nan")
    tag = f" vdim={vdim}{'v' if byvdim else 'n'}" if vdim else ""
    ndofs = ndofs*max(1, vdim)
    print(f"{problem}{tag} L{layout} p={p} n={n} ne={ne} ndofs={ndofs} | L2E {t1*1e6:8.1f} us {b1/t1/1e9:7.1f} GB/s "
          f"({b1/t1/8e12*100:4.1f}%) | E2L {t2*1e6:8.1f} us {b2/t2/1e9:7.1f} GB/s | apply {ndofs/(t1+t2)/1e9:6.2f} GDOF/s "
          f"{(b1+b2)/(t1+t2)/8e12*100:4.1f}% of 8TB/s", flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--problem", default="bp3")
    ap.add_argument("--p", default="4")
    ap.add_argument("--dofs", type=float, default=1e7)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--which", default="both")
    ap.add_argument("--layout", default="1")
    ap.add_argument("--vdim", type=int, default=0)
    ap.add_argument("--byvdim", type=int, default=1)
    a = ap.parse_args()
    import torch
    from mfem_amd import capi
    torch.cuda.init()
    capi.call("mfem_amd_init", 0)
    for prob in a.problem.split(","):
        for p in [int(s) for s in a.p.split(",")]:
            n = max(1, round((a.dofs/max(1, a.vdim))**(1/3)/p))
            for lay in [int(v) for v in a.layout.split(',')]:
                run(prob, p, n, a.steps, a.which, lay, a.vdim, a.byvdim)
