import sys, pathlib, ctypes as C
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent)); sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent/"tests"))
import numpy as np, torch
from conftest import golden_cases, load_case
import gpu_helpers as gh
from mfem_amd import capi
capi.call("mfem_amd_init", 0)
for name in golden_cases():
    c = load_case(name)
    D, NE, n = c["D1D"], c["NE"], c["ndofs"]
    if D < 3: continue
    for layout in (0, 1):
        if layout and not (capi.lib.mfem_amd_has_specialization(c["kind"], D, c["Q1D"]) and NE % 2 == 0): continue
        op = gh.DeviceOperator(c, c["ess_tdof"], layout=layout)
        x = gh.dev(c["x"])
        y0 = gh.host(op.constrained_mult(x))
        op.op.direct_rows = NE*(D-2)**3
        y1 = gh.host(op.constrained_mult(x))
        bad = np.flatnonzero(y0 != y1)
        i0 = n - NE*(D-2)**3
        if len(bad):
            li = (bad - i0) % ((D-2)**3)
            print(name, "layout", layout, "D", D, "Q", c["Q1D"], "NE", NE, "bad", len(bad), "of", NE*(D-2)**3, "rows-i0", (bad - i0)[:12], "local", li[:12], "elems", ((bad-i0)//((D-2)**3))[:12])
        else:
            print(name, "layout", layout, "ok")
