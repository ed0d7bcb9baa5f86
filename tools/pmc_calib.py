#!/usr/bin/env python3
"""Counter calibration (tuning aid): kernels of KNOWN byte counts in this library's own access widths, then the two
kernels of the apply, so that the memory-side request counters can be read as bytes (MI355X_MICROARCH.md: only the
16-B-per-lane streaming read is calibrated there).  Run under rocprofv3 --pmc; tools/pmc_summary.py averages per kernel.
  identity gather  n entries: reads 4n (int32, 4 B/lane) + 8n (double via the map, 8 B/lane), writes 8n
  vec_dot          n entries: reads 16n (two doubles streams), writes ~0
"""
import ctypes as C
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import torch
from mfem_amd import capi
import kbench

torch.cuda.init()
capi.call("mfem_amd_init", 0)
dev = torch.device("cuda", 0)
n = 1 << 25
ptr = lambda t: C.c_void_p(t.data_ptr())
gm = torch.arange(n, dtype=torch.int32, device=dev)
x = torch.rand(n, dtype=torch.float64, device=dev)
y = torch.empty_like(x)
ws = torch.zeros(capi.lib.mfem_amd_reduce_workspace_bytes(), dtype=torch.uint8, device=dev)
r = torch.zeros(1, dtype=torch.float64, device=dev)
for _ in range(8):
    capi.call("mfem_amd_restrict_gather", n, ptr(gm), ptr(x), ptr(y), None)
    capi.call("mfem_amd_vec_dot", n, ptr(x), ptr(y), ptr(r), ptr(ws), None)
torch.cuda.synchronize()
print(f"calib n={n}: gather reads {12*n} writes {8*n}; dot reads {16*n}", flush=True)
del gm, x, y
for prob in sys.argv[1:] or ["bp3", "bp1"]:
    kbench.run(prob, 4, 54, 5, "both")
