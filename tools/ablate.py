#!/usr/bin/env python3
"""Tuning aid: time the ablation variants of the BP3 element kernel (tools/ablate.hip)."""
import ctypes as C, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench, torch
from mfem_amd import capi
from mfem_amd.cartesian import CartesianProblem
lib = C.CDLL(str(ROOT / "tools" / "libablate.so"))
hd = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
lib.ablate_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, hd, hd, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
dev = torch.device("cuda", 0)
for p in [int(a) for a in sys.argv[1].split(",")]:
    D, Q = p + 1, p + 2
    n = round(1e7**(1/3)/p)
    prob = CartesianProblem(p, n, n, n)
    ne, ndofs = prob.ne, prob.ndofs
    from mfem_amd.tables import bakeoff_tables
    _, _, _, B, G, _, _, _ = bakeoff_tables('bp3', p)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    gm = t(prob.gather_map)
    pa = torch.rand(capi.lib.mfem_amd_pa_data_size(1, 1, Q, ne), dtype=torch.float64, device=dev)
    x = torch.rand(ndofs, dtype=torch.float64, device=dev)
    yE = torch.empty(ne*D**3, dtype=torch.float64, device=dev)
    b1, _ = bench.algorithmic_bytes(1, D, Q, ne, ndofs)
    for gpc in [int(a) for a in sys.argv[2].split(",")]:
        for abl, blk in ((0, 1), (256, 1), (17, 1), (273, 1), (5, 1)):
            def f():
                rc = lib.ablate_run(D, abl, blk, ne, B, G, pa.data_ptr(), x.data_ptr(), gm.data_ptr(), yE.data_ptr(), gpc)
                assert rc == 0, rc
            for _ in range(3): f()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): f()
            e1.record(); torch.cuda.synchronize()
            tt = e0.elapsed_time(e1)*1e-3/10
            print(f"p={p} grid/CU={gpc} blk={blk} abl={abl} ({ {0:'full',1:'mem-only',2:'compute-only',5:'mem-only no-store',17:'mem-only coalesced-store',273:'mem-only coalesced-store, contiguous ranges',256:'full, contiguous ranges',145:'mem-only wave0-only store',49:'mem-only coalesced nt-store',81:'mem-only store to small L2-resident region',9:'mem-only no-gather',13:'mem-only no-store no-gather'}[abl]}): {tt*1e6:8.1f} us  {b1/tt/1e9:7.1f} GB/s", flush=True)
