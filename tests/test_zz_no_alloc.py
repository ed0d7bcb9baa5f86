"""mfem_amd_vec_dot_noalloc (library-owned scratch, no allocation) against mfem_amd_vec_dot, through the C ABI:
bit-identical results on two calls in a row, the second with a smaller input."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_vec_dot_noalloc_matches_vec_dot():
    import torch
    from mfem_amd import capi
    assert torch.cuda.is_available()
    capi.call("mfem_amd_init", 0)
    dev = torch.device("cuda", 0)
    ws = torch.empty(capi.lib.mfem_amd_reduce_workspace_bytes(), dtype=torch.uint8, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    g = torch.Generator(device="cpu"); g.manual_seed(7)
    for n in (3_000_017, 1_234, 9_000_001, 1, 0):          # the second call is smaller than the first
        x = (torch.rand(max(n, 1), dtype=torch.float64, generator=g) - 0.5).to(dev)[:n]
        y = (torch.rand(max(n, 1), dtype=torch.float64, generator=g) - 0.5).to(dev)[:n]
        ref = torch.full((1,), 123.0, dtype=torch.float64, device=dev)
        out = torch.full((1,), -321.0, dtype=torch.float64, device=dev)
        capi.call("mfem_amd_vec_dot", n, p(x), p(y), p(ref), p(ws), None)
        capi.call("mfem_amd_vec_dot_noalloc", n, p(x), p(y), p(out), None)
        torch.cuda.synchronize()
        a, b = ref.cpu().numpy(), out.cpu().numpy()
        assert a.tobytes() == b.tobytes(), (n, a, b)                      # bit-identical
        exact = float(np.dot(x.cpu().numpy().astype(np.longdouble), y.cpu().numpy().astype(np.longdouble)))
        assert abs(b[0] - exact) <= 1e-12*max(1.0, float(np.abs(x.cpu().numpy()*y.cpu().numpy()).sum()))


def test_vec_dot_noalloc_rejects_oversize():
    import torch
    from mfem_amd import capi
    capi.call("mfem_amd_init", 0)
    one = torch.zeros(1, dtype=torch.float64, device="cuda")
    rc = capi.lib.mfem_amd_vec_dot_noalloc((1 << 33) + 1, C.c_void_p(one.data_ptr()), C.c_void_p(one.data_ptr()),
                                           C.c_void_p(one.data_ptr()), None)
    assert rc != 0 and b"exceeds the maximum" in capi.lib.mfem_amd_last_error()
