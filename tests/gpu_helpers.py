"""Helpers for the -m gpu parity tests: device buffers via torch, calls through the C ABI."""
import ctypes as C

import numpy as np
import torch

from mfem_amd import capi

DEV = "cuda:0"


def dev(a, dtype=None):
    a = np.ascontiguousarray(a, dtype=dtype)
    return torch.from_numpy(a).to(DEV)


def zeros(n, dtype=torch.float64):
    return torch.zeros(int(n), dtype=dtype, device=DEV)


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def host(t):
    torch.cuda.synchronize()
    return t.cpu().numpy()


def hB(a):
    return np.ascontiguousarray(a, np.float64)


class DeviceOperator:
    """mfem_amd_pa_operator filled from a golden case (or any dict with the same keys)."""

    def __init__(self, c, ess=None, layout=0):
        self.c = c
        self.layout = layout
        D1, NE = c["D1D"], c["NE"]
        self.B, self.G = hB(c["B"]), hB(c["G"])
        self.pa = dev(c["pa_data"])
        if layout:
            nb = capi.lib.mfem_amd_pa_data_size(c["kind"], 1, c["Q1D"], NE)
            blk = zeros(nb)
            capi.call("mfem_amd_pa_data_to_blocked", c["kind"], c["Q1D"], NE, ptr(self.pa), ptr(blk), None)
            self.pa = blk
        self.gm = dev(c["gather_map"], np.int32)
        self.off = dev(c["offsets"], np.int32)
        self.ind = dev(c["indices"], np.int32)
        self.yE = zeros(NE*D1**3)
        ess = np.asarray(ess if ess is not None else [], np.int32)
        self.ess = dev(ess, np.int32) if len(ess) else None
        self.marker = zeros(c["ndofs"], torch.uint8)
        self.masked = None
        if len(ess):
            capi.call("mfem_amd_set_marker", len(ess), ptr(self.ess), ptr(self.marker), None)
            self.masked = zeros(NE*D1**3, torch.int32)
            capi.call("mfem_amd_mask_gather_map", NE*D1**3, ptr(self.gm), ptr(self.marker), ptr(self.masked), None)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        self.op = capi.PaOperator(c["kind"], layout, D1, c["Q1D"], NE, c["ndofs"], vp(self.B), vp(self.G),
                                  ptr(self.pa), ptr(self.gm), ptr(self.masked), ptr(self.off), ptr(self.ind),
                                  len(ess), ptr(self.ess), ptr(self.marker), ptr(self.yE))

    def apply_L2L(self, x):
        """unconstrained y = A x: fused gather+apply, then transpose-gather"""
        c = self.c
        y = zeros(c["ndofs"])
        capi.call("mfem_amd_pa_apply_L2E", c["kind"], self.layout, c["D1D"], c["Q1D"], c["NE"], self.B, self.G, ptr(self.pa),
                  ptr(self.gm), ptr(x), ptr(self.yE), None)
        capi.call("mfem_amd_restrict_scatter", c["ndofs"], ptr(self.off), ptr(self.ind), ptr(self.yE), ptr(y), 0, 0, None)
        return y

    def constrained_mult(self, x):
        y = zeros(self.c["ndofs"])
        capi.call("mfem_amd_pa_constrained_mult", C.byref(self.op), ptr(x), ptr(y), None)
        return y

    def pcg(self, b, dinv, rel_tol, abs_tol=0.0, max_iter=2000, check_every=1):
        n = self.c["ndofs"]
        ws = zeros(capi.lib.mfem_amd_pcg_workspace_bytes(n, max_iter), torch.uint8)
        x = zeros(n)
        res = capi.PcgResult()
        hist = np.zeros(max_iter + 2)
        capi.call("mfem_amd_pcg_pa", C.byref(self.op), ptr(dinv), ptr(b), ptr(x), rel_tol, abs_tol, max_iter,
                  check_every, ptr(ws), C.byref(res), hist.ctypes.data_as(C.c_void_p), None)
        return x, res, hist[:res.history_count]
