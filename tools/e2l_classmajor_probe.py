#!/usr/bin/env python3
"""Probe (tuning aid): E->L over a CLASS-MAJOR E-vector -- the interior entries of all elements first (element-major),
then the face, edge and vertex entries -- with the shipped kernel and remapped indices.  In the reference layout a
128-B line of an element's block holds entries of all four classes, which MFEM's numbering visits in four separate
sweeps (1.66x over-fetch, bench_traffic.json); class-major, each sweep streams its own segment.  y is bit-identical.
This measures the E->L side only; the element kernel would have to store through the same map."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def class_major_map(p, ne):
    """position of E-entry (e, local) in the class-major layout"""
    D = p + 1
    k, j, i = np.meshgrid(np.arange(D), np.arange(D), np.arange(D), indexing="ij")    # local = i + D*(j + D*k)
    onb = [(c == 0) | (c == p) for c in (i, j, k)]
    nb = onb[0].astype(int) + onb[1] + onb[2]
    cls = nb.reshape(-1)                                           # 0 interior, 1 face, 2 edge, 3 vertex
    # sub-entity id inside the class (face id / edge id), then lexicographic: entries of one face stay together
    fid = np.where(onb[0], np.where(i == 0, 0, 1), np.where(onb[1], np.where(j == 0, 2, 3), np.where(k == 0, 4, 5)))
    eid = (np.where(~onb[0], 0, np.where(~onb[1], 1, 2))*4 +
           np.where(~onb[0], (j == p)*1 + (k == p)*2, np.where(~onb[1], (i == p)*1 + (k == p)*2, (i == p)*1 + (j == p)*2)))
    sub = np.where(nb == 1, fid, np.where(nb == 2, eid, 0)).reshape(-1)
    order = np.lexsort((np.arange(D**3), sub, cls))                # local indices sorted by (class, sub-entity, lexicographic)
    counts = np.bincount(cls, minlength=4)
    rank_in_class = np.zeros(D**3, np.int64)
    start = 0
    for c in range(4):
        loc = order[start:start + counts[c]]
        rank_in_class[loc] = np.arange(counts[c])
        start += counts[c]
    seg_base = np.concatenate(([0], np.cumsum(counts*ne)))[:4]
    e = np.arange(ne, dtype=np.int64)[:, None]
    pos = seg_base[cls][None, :] + e*counts[cls][None, :] + rank_in_class[None, :]
    return pos.reshape(-1)                                         # old E index -> new E index


def run(p, dofs):
    import torch
    from mfem_amd import capi
    from mfem_amd.cartesian import CartesianProblem
    dev = torch.device("cuda", 0)
    n = max(1, round(dofs**(1/3)/p))
    prob = CartesianProblem(p, n, n, n)
    D3, ne, nd = (p + 1)**3, prob.ne, prob.ndofs
    pos = class_major_map(p, ne)
    assert np.array_equal(np.sort(pos), np.arange(ne*D3))
    t = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dt)).to(dev)
    ptr = lambda x: C.c_void_p(x.data_ptr())
    xE = torch.rand(ne*D3, dtype=torch.float64, device=dev)
    d_pos = t(pos, np.int64)
    xE_cm = torch.empty_like(xE); xE_cm[d_pos] = xE
    ind_cm = pos[prob.indices.astype(np.int64)]
    d_off, d_ind, d_ind_cm = t(prob.offsets, np.int32), t(prob.indices, np.int32), t(ind_cm, np.int32)
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

    f0 = lambda: capi.call("mfem_amd_restrict_scatter", nd, ptr(d_off), ptr(d_ind), ptr(xE), ptr(y0), 0, 0, None)
    f1 = lambda: capi.call("mfem_amd_restrict_scatter", nd, ptr(d_off), ptr(d_ind_cm), ptr(xE_cm), ptr(y1), 0, 0, None)
    t0, t1 = timed(f0), timed(f1)
    print(f"p={p} n={n} ndofs={nd}: reference layout {t0:6.1f} us | class-major {t1:6.1f} us (bit-identical {bool(torch.equal(y0, y1))})", flush=True)


if __name__ == "__main__":
    import torch
    from mfem_amd import capi
    torch.cuda.init()
    capi.call("mfem_amd_init", 0)
    for p in [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "2,3,4,5,6,8").split(",")]:
        run(p, 1e7)
