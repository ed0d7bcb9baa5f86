#!/usr/bin/env python3
"""Tuning aid: time the compute-phase ablations of the element kernel (tools/ablate2.hip)."""
import ctypes as C, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench, torch
from mfem_amd import capi
from mfem_amd.cartesian import CartesianProblem
from mfem_amd.tables import bakeoff_tables
torch.cuda.init()
capi.call("mfem_amd_init", 0)
lib = C.CDLL(str(ROOT / "tools" / "libablate2.so"))
hd = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
lib.ablate2_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, hd, hd, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
dev = torch.device("cuda", 0)
NAMES = {0: "full kernel", 2: "contractions only (no pa_data)", 514: "  + literal basis (no scalar loads)", 1026: "  - z loop",
         2050: "  - y contraction", 4098: "  - transposed y", 7170: "  - z, y, transposed y (x phases + LDS staging only)"}
for p in [int(a) for a in sys.argv[1].split(",")]:
    D, Q = p + 1, p + 2
    n = round(1e7**(1/3)/p)
    prob = CartesianProblem(p, n, n, n)
    ne, ndofs = prob.ne, prob.ndofs
    _, _, _, B, G, _, _, _ = bakeoff_tables("bp3", p)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    gm = t(prob.gather_map)
    pa = torch.rand(capi.lib.mfem_amd_pa_data_size(1, 1, Q, ne), dtype=torch.float64, device=dev)
    x = torch.rand(ndofs, dtype=torch.float64, device=dev)
    yE = torch.empty(ne*D**3, dtype=torch.float64, device=dev)
    for abl, rb in ((0, 1), (2, 1), (514, 1), (1026, 1), (2050, 1), (4098, 1), (7170, 1), (2, 2), (514, 2), (2, 0)):
        def f():
            rc = lib.ablate2_run(D, abl, rb, ne, B, G, pa.data_ptr(), x.data_ptr(), gm.data_ptr(), yE.data_ptr())
            assert rc == 0, rc
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): f()
        e1.record(); torch.cuda.synchronize()
        tt = e0.elapsed_time(e1)*1e-3/10
        print(f"p={p} RB={rb} abl={abl:5d} {NAMES[abl]:60s} {tt*1e6:8.1f} us  {tt/ne*1e9:6.2f} ns/element", flush=True)
