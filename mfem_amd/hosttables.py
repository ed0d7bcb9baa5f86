"""The reference's own numbering for the synthetic boxes: tables built by the C++ host layer (libmfem_amd_cxx.so,
include/mfem_amd_host.h) -- Mesh::MakeCartesian3D in its space-filling-curve element order, H1 space numbered
vertex / edge / face / interior as fem/fespace.cpp numbers it -- so that the Python drivers time what the reference's
callers (examples/ex1.cpp, tests/benchmarks/bench_ceed.cpp) produce.  Host-only integer plumbing."""
import ctypes as C
from pathlib import Path

import numpy as np

from .cartesian import CartesianProblem

_SO = Path(__file__).resolve().parent / "libmfem_amd_cxx.so"
_lib = None


class _Tables(C.Structure):
    """struct mfem_amd_host_tables"""
    _fields_ = [("ne", C.c_int64), ("ndofs", C.c_int64), ("ness", C.c_int64), ("D1D", C.c_int),
                ("gather_map", C.POINTER(C.c_int)), ("nodes_E", C.POINTER(C.c_double)), ("ess", C.POINTER(C.c_int)),
                ("owner", C.c_void_p)]


def _load():
    global _lib
    if _lib is None:
        if not _SO.exists():
            raise ImportError(f"{_SO} is missing: build it with make -C mfem_amd/host")
        from . import capi  # noqa: F401  (loads libmfem_amd.so, which the host library links, from the same directory)
        _lib = C.CDLL(str(_SO))
        _lib.mfem_amd_host_cartesian_tables.restype = C.c_int
        _lib.mfem_amd_host_cartesian_tables.argtypes = [C.c_int]*4 + [C.c_double]*6 + [C.c_int, C.POINTER(_Tables)]
        _lib.mfem_amd_host_tables_free.restype = None
        _lib.mfem_amd_host_tables_free.argtypes = [C.POINTER(_Tables)]
    return _lib


class MfemBox:
    """CartesianProblem's interface (gather_map, nodes_E, boundary_dofs, n1, ...) on MFEM's numbering and element
    order.  self.perm[grid id] = L-dof id here; self.element_order[e] = lexicographic index of element e."""

    def __init__(self, p, nx, ny, nz, sx=1.0, sy=1.0, sz=1.0, origin=(0, 0, 0), global_n=None, transpose=False):
        grid = CartesianProblem(p, nx, ny, nz, sx, sy, sz, origin=origin, global_n=global_n, transpose=False)
        self._grid = grid
        self.p, self.n, self.n1, self.ne, self.ndofs = p, grid.n, grid.n1, grid.ne, grid.ndofs
        self.origin, self.global_n = grid.origin, grid.global_n
        gn = grid.global_n
        h = (sx/gn[0], sy/gn[1], sz/gn[2])
        o = tuple(origin[d]*h[d] for d in range(3))
        t = _Tables()
        lib = _load()
        rc = lib.mfem_amd_host_cartesian_tables(p, nx, ny, nz, nx*h[0], ny*h[1], nz*h[2], o[0], o[1], o[2], 1, C.byref(t))
        if rc != 0:
            raise RuntimeError("mfem_amd_host_cartesian_tables failed")
        try:
            D3 = (p + 1)**3
            assert (t.ne, t.ndofs, t.D1D) == (grid.ne, grid.ndofs, p + 1)
            gm = np.ctypeslib.as_array(t.gather_map, shape=(t.ne*D3,)).copy().reshape(t.ne, D3)
            nodes = np.ctypeslib.as_array(t.nodes_E, shape=(t.ne*24,)).copy()
        finally:
            lib.mfem_amd_host_tables_free(C.byref(t))
        c0 = nodes.reshape(self.ne, 3, 8)[:, :, 0]
        ex, ey, ez = (np.rint((c0[:, d] - o[d])/h[d]).astype(np.int64) for d in range(3))
        self.element_order = ex + nx*(ey + ny*ez)
        gg = grid.gather_map.reshape(self.ne, D3)[self.element_order]
        perm = np.full(self.ndofs, -1, np.int64)
        perm[gg.reshape(-1)] = gm.reshape(-1)
        assert perm.min() >= 0 and np.array_equal(perm[gg], gm), "the grid and the reference numbering disagree"
        self.perm = perm
        self.gather_map = np.ascontiguousarray(gm.reshape(-1), np.int32)
        self.nodes_E = nodes
        self.offsets = self.indices = None
        if transpose:
            self.indices = np.argsort(self.gather_map, kind="stable").astype(np.int32)
            self.offsets = np.zeros(self.ndofs + 1, np.int32)
            np.cumsum(np.bincount(self.gather_map, minlength=self.ndofs), out=self.offsets[1:])

    def boundary_dofs(self):
        """essential dofs: the faces of the GLOBAL box (CartesianProblem.boundary_dofs), in this numbering"""
        return np.sort(self.perm[self._grid.boundary_dofs()]).astype(np.int32)
