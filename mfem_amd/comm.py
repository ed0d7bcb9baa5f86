"""ctypes binding of include/mfem_amd_comm.h (libmfem_amd_comm.so): the process group and the shared-dof exchange
written in C++ on RCCL.  With it the callbacks of ``struct mfem_amd_comm`` are C functions: a PCG iteration of
``mfem_amd_pcg_pa_par`` runs without the interpreter (the torch.distributed transport of mfem_amd/parallel.py calls
back into Python four times per iteration).  The shared-dof TABLES are the ones mfem_amd/parallel.py builds."""
import ctypes as C
import os
from pathlib import Path

import numpy as np
import torch

from . import capi

_SO = Path(__file__).resolve().parent / "libmfem_amd_comm.so"
if not _SO.exists():
    raise ImportError(f"{_SO} is missing: build it with make -C mfem_amd/csrc")
lib = C.CDLL(str(_SO))

vp, cint, i64 = C.c_void_p, C.c_int, C.c_int64
_ip = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_SIGS = {
    "mfem_amd_comm_last_error": (C.c_char_p, []),
    "mfem_amd_world_create": (cint, [cint, cint, C.c_char_p, cint, cint, cint, C.POINTER(vp)]),
    "mfem_amd_world_rank": (cint, [vp]),
    "mfem_amd_world_size": (cint, [vp]),
    "mfem_amd_world_transport": (cint, [vp]),
    "mfem_amd_world_allreduce_host": (cint, [vp, np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS"), cint, cint]),
    "mfem_amd_world_barrier": (cint, [vp]),
    "mfem_amd_world_destroy": (cint, [vp]),
    "mfem_amd_exchange_create": (cint, [vp, cint, _ip, _ip, cint, _ip, _ip, vp, vp, i64, vp, vp, vp, C.POINTER(vp),
                                        C.POINTER(capi.Comm)]),
    "mfem_amd_exchange_destroy": (cint, [vp]),
}
EXPORTS = sorted(_SIGS)
for _n, (_r, _a) in _SIGS.items():
    _f = getattr(lib, _n)
    _f.restype, _f.argtypes = _r, _a

TRANSPORT_RCCL, TRANSPORT_TCP = 0, 1


class CommError(RuntimeError):
    pass


def _check(rc):
    if rc != 0:
        raise CommError(lib.mfem_amd_comm_last_error().decode())


class World:
    """The process group (Mpi::Init's stand-in).  Arguments default to the launcher's environment (RANK,
    WORLD_SIZE, MASTER_ADDR, MASTER_PORT); `port_offset` keeps clear of torch.distributed's own store."""

    def __init__(self, rank=-1, size=1, addr="127.0.0.1", port=29400, port_offset=23, transport=-1):
        self._h = vp()
        _check(lib.mfem_amd_world_create(rank, size, addr.encode(), port, port_offset, transport, C.byref(self._h)))
        self.rank, self.size = lib.mfem_amd_world_rank(self._h), lib.mfem_amd_world_size(self._h)
        self.transport = lib.mfem_amd_world_transport(self._h)

    def allreduce_host(self, values, op="sum"):
        a = np.ascontiguousarray(np.atleast_1d(np.asarray(values, np.float64))).copy()
        _check(lib.mfem_amd_world_allreduce_host(self._h, a, a.size, 1 if op == "max" else 0))
        return a

    def barrier(self):
        _check(lib.mfem_amd_world_barrier(self._h))

    def close(self):
        if self._h:
            lib.mfem_amd_world_destroy(self._h)
            self._h = vp()


class NativeExchange:
    """P / P^T / all-reduce of one space through the C++ exchange.  Same interface as parallel.Exchange where the
    drivers use it (forward, reverse, make_comm); every kernel and RCCL call is ordered on the stream handed over."""

    def __init__(self, world, part, device, vdim=0):
        from .parallel import _VdofView
        if vdim:
            part = _VdofView(part, vdim)
        self.world, self.part, self.device = world, part, torch.device(device)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.int32)).to(self.device)
        self.shr_ltdof, self.ext_ldof = t(part.shr_ltdof), t(part.ext_ldof)
        self.unq_ltdof, self.unq_shr_i, self.unq_shr_j = t(part.unq_ltdof), t(part.unq_shr_i), t(part.unq_shr_j)
        ia = lambda v: np.ascontiguousarray(v, np.int32)
        upr, ups = ia([r for r, _ in part.upper]), ia(part.shr_sizes)
        lor, los = ia([r for r, _ in part.lower]), ia(part.ext_sizes)
        self._h, self.comm = vp(), capi.Comm()
        p = lambda x: vp(x.data_ptr()) if x.numel() else None
        _check(lib.mfem_amd_exchange_create(world._h, len(upr), upr, ups, len(lor), lor, los, p(self.shr_ltdof), p(self.ext_ldof),
                                            self.unq_ltdof.numel(), p(self.unq_ltdof), p(self.unq_shr_i), p(self.unq_shr_j),
                                            C.byref(self._h), C.byref(self.comm)))
        self.error = None

    def _call(self, fn, v, stream):
        ptr = v if isinstance(v, int) else v.data_ptr()
        s = stream.value if isinstance(stream, C.c_void_p) else stream
        rc = fn(self.comm.ctx, vp(ptr), vp(s or 0))
        _check(rc)

    def forward(self, v, stream=None):
        self._call(self.comm.forward, v, stream)

    def reverse(self, v, stream=None):
        self._call(self.comm.reverse, v, stream)

    def allreduce_sum(self, t, stream=None):
        s = stream.value if isinstance(stream, C.c_void_p) else stream
        _check(self.comm.allreduce_sum(self.comm.ctx, vp(t.data_ptr()), int(t.numel()), vp(s or 0)))

    def make_comm(self, workspace):
        return self.comm

    def close(self):
        if self._h:
            lib.mfem_amd_exchange_destroy(self._h)
            self._h = vp()
