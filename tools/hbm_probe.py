#!/usr/bin/env python3
"""Calibrate achievable HBM rates on the box with plain torch ops (read-only, copy, write)
and with the library's own streaming kernels."""
import ctypes as C, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from mfem_amd import capi
dev = torch.device("cuda", 0)
n = 200_000_000   # 1.6 GB of doubles
a = torch.rand(n, dtype=torch.float64, device=dev)
b = torch.empty_like(a)
def timed(f, k=10):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e-3/k
t = timed(lambda: torch.sum(a));            print(f"torch.sum   (read  {n*8/1e9:.1f} GB): {n*8/t/1e12:.2f} TB/s")
t = timed(lambda: b.copy_(a));              print(f"copy_       (r+w   {2*n*8/1e9:.1f} GB): {2*n*8/t/1e12:.2f} TB/s")
t = timed(lambda: b.fill_(1.0));            print(f"fill_       (write {n*8/1e9:.1f} GB): {n*8/t/1e12:.2f} TB/s")
a32 = a.view(torch.float32)
t = timed(lambda: torch.sum(a32));          print(f"torch.sum f32 (read): {n*8/t/1e12:.2f} TB/s")
ws = torch.empty(capi.lib.mfem_amd_reduce_workspace_bytes(), dtype=torch.uint8, device=dev)
r = torch.empty(1, dtype=torch.float64, device=dev)
p = lambda x: C.c_void_p(x.data_ptr())
t = timed(lambda: capi.call("mfem_amd_vec_dot", n, p(a), p(a), p(r), p(ws), None)); print(f"vec_dot(a,a) (read once): {n*8/t/1e12:.2f} TB/s")
t = timed(lambda: capi.call("mfem_amd_vec_dot", n//2, p(a), p(b), p(r), p(ws), None)); print(f"vec_dot(a,b) (2 streams): {n*8/t/1e12:.2f} TB/s")
t = timed(lambda: capi.call("mfem_amd_vec_axpby", n, 1.0, p(a), 0.0, None, p(b), None)); print(f"axpby copy (r+w): {2*n*8/t/1e12:.2f} TB/s")
