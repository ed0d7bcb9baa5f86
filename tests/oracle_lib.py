"""ctypes binding of the CPU oracle (oracle/pa_oracle.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
_SO = ROOT / "oracle" / "libpa_oracle.so"


def _load():
    src = ROOT / "oracle" / "pa_oracle.c"
    if not _SO.exists() or _SO.stat().st_mtime < src.stat().st_mtime:
        subprocess.check_call(["make", "-C", str(ROOT / "oracle")], stdout=subprocess.DEVNULL)
    return C.CDLL(str(_SO))


lib = _load()
_d = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_i = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
i64 = C.c_int64


class OrcOp(C.Structure):
    _fields_ = [("kind", C.c_int), ("D1D", C.c_int), ("Q1D", C.c_int),
                ("ne", i64), ("ndofs", i64),
                ("B", C.c_void_p), ("G", C.c_void_p), ("pa_data", C.c_void_p),
                ("gather_map", C.c_void_p), ("offsets", C.c_void_p), ("indices", C.c_void_p),
                ("ness", C.c_int), ("ess", C.c_void_p),
                ("xE", C.c_void_p), ("yE", C.c_void_p), ("w", C.c_void_p), ("z", C.c_void_p),
                ("vdim", C.c_int), ("byvdim", C.c_int), ("coeff_vdim", C.c_int)]


def _sig(name, res, *args):
    f = getattr(lib, name)
    f.restype = res
    f.argtypes = list(args)
    return f


_sig("orc_gauss_legendre", None, C.c_int, _d, _d)
_sig("orc_gauss_lobatto", None, C.c_int, _d, _d)
_sig("orc_tensor_weights_3d", None, C.c_int, _d, _d)
_sig("orc_basis_1d", None, C.c_int, _d, C.c_int, _d, _d, _d)
_sig("orc_restriction_build", None, C.c_int, C.c_int, C.c_int, _i, C.c_void_p, _i, _i, _i)
_sig("orc_restriction_mult", None, i64, C.c_int, C.c_int, C.c_int, i64, _i, _d, _d)
_sig("orc_restriction_mult_transpose", None, i64, _i, _i, _d, _d, C.c_int, C.c_int)
_sig("orc_jacobians_3d", None, C.c_int, C.c_int, i64, _d, _d, _d, _d)
_sig("orc_det_3d", None, C.c_int, i64, _d, _d)
_sig("orc_diffusion_setup_3d", None, C.c_int, C.c_int, i64, _d, _d, _d, C.c_int, _d)
_sig("orc_mass_setup_3d", None, C.c_int, i64, _d, _d, _d, C.c_int, _d)
_sig("orc_diffusion_apply_3d", None, C.c_int, C.c_int, i64, _d, _d, _d, _d, _d)
_sig("orc_mass_apply_3d", None, C.c_int, C.c_int, i64, _d, _d, _d, _d)
_sig("orc_diffusion_diag_3d", None, C.c_int, C.c_int, i64, _d, _d, _d, _d)
_sig("orc_diffusion_apply_sym_3d", None, C.c_int, C.c_int, i64, C.c_int, _d, _d, _d, _d, _d)
_sig("orc_diffusion_diag_sym_3d", None, C.c_int, C.c_int, i64, C.c_int, _d, _d, _d, _d)
_sig("orc_mass_diag_3d", None, C.c_int, C.c_int, i64, _d, _d, _d)
_sig("orc_domain_lf_3d", None, C.c_int, C.c_int, i64, _d, _d, _d, C.c_double, _d)
_sig("orc_restriction_mult_transpose_vdim", None, i64, C.c_int, C.c_int, C.c_int, _i, _i, _d, _d, C.c_int, C.c_int)
_sig("orc_vector_diffusion_apply_3d", None, C.c_int, C.c_int, i64, C.c_int, _d, _d, _d, _d, _d)
_sig("orc_vector_mass_apply_3d", None, C.c_int, C.c_int, i64, C.c_int, C.c_int, _d, _d, _d, _d)
_sig("orc_pa_mult", None, C.POINTER(OrcOp), _d, _d)
_sig("orc_pa_diag", None, C.POINTER(OrcOp), _d)
_sig("orc_constrained_mult", None, C.POINTER(OrcOp), _d, _d)
_sig("orc_eliminate_rhs", None, C.POINTER(OrcOp), _d, _d)
_sig("orc_jacobi_setup", None, i64, _d, C.c_int, _i, _d)
_sig("orc_pcg", C.c_int, C.POINTER(OrcOp), C.c_void_p, _d, _d, C.c_double, C.c_double, C.c_int,
     _d, C.POINTER(C.c_int))
_sig("orc_num_threads", C.c_int)
_sig("orc_set_num_threads", None, C.c_int)
_sig("orc_set_threaded_vectors", None, C.c_int)       # timing leg of bench.py only


def gauss_legendre(n):
    x, w = np.zeros(n), np.zeros(n)
    lib.orc_gauss_legendre(n, x, w)
    return x, w


def gauss_lobatto(n):
    x, w = np.zeros(n), np.zeros(n)
    lib.orc_gauss_lobatto(n, x, w)
    return x, w


def tensor_weights(w1):
    Q = len(w1)
    W = np.zeros(Q**3)
    lib.orc_tensor_weights_3d(Q, np.ascontiguousarray(w1), W)
    return W


def basis_1d(p, qpts):
    """H1 GLL basis of order p tabulated at qpts -> B, G in MFEM layout [q + Q*d]."""
    nodes, _ = gauss_lobatto(p + 1)
    Q = len(qpts)
    B, G = np.zeros(Q*(p + 1)), np.zeros(Q*(p + 1))
    lib.orc_basis_1d(p, nodes, Q, np.ascontiguousarray(qpts), B, G)
    return B, G


def restriction_build(ne, dof, ndofs, elem2dof, dof_map):
    g = np.zeros(ne*dof, np.int32); off = np.zeros(ndofs + 1, np.int32); ind = np.zeros(ne*dof, np.int32)
    dm = np.ascontiguousarray(dof_map, np.int32)
    lib.orc_restriction_build(ne, dof, ndofs, np.ascontiguousarray(elem2dof, np.int32),
                              dm.ctypes.data_as(C.c_void_p), g, off, ind)
    return g, off, ind


class Operator:
    """L->L PA operator on the oracle (keeps numpy arrays alive)."""

    def __init__(self, kind, D1D, Q1D, ne, ndofs, B, G, pa_data, gather_map, offsets, indices, ess=None,
                 vdim=0, byvdim=0, coeff_vdim=1):
        vd = max(1, vdim)
        self.keep = dict(B=np.ascontiguousarray(B, np.float64), G=np.ascontiguousarray(G, np.float64),
                         pa=np.ascontiguousarray(pa_data, np.float64),
                         g=np.ascontiguousarray(gather_map, np.int32),
                         o=np.ascontiguousarray(offsets, np.int32),
                         i=np.ascontiguousarray(indices, np.int32),
                         ess=np.ascontiguousarray(ess if ess is not None else [], np.int32),
                         xE=np.zeros(ne*D1D**3*vd), yE=np.zeros(ne*D1D**3*vd), w=np.zeros(ndofs*vd), z=np.zeros(ndofs*vd))
        k = self.keep
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        self.op = OrcOp(kind, D1D, Q1D, ne, ndofs, p(k["B"]), p(k["G"]), p(k["pa"]), p(k["g"]), p(k["o"]),
                        p(k["i"]), len(k["ess"]), p(k["ess"]), p(k["xE"]), p(k["yE"]), p(k["w"]), p(k["z"]),
                        vdim, byvdim, coeff_vdim)
        self.n = ndofs*vd

    def mult(self, x):
        y = np.zeros(self.n); lib.orc_pa_mult(C.byref(self.op), np.ascontiguousarray(x), y); return y

    def diag(self):
        y = np.zeros(self.n); lib.orc_pa_diag(C.byref(self.op), y); return y

    def constrained_mult(self, x):
        y = np.zeros(self.n); lib.orc_constrained_mult(C.byref(self.op), np.ascontiguousarray(x), y); return y

    def eliminate_rhs(self, x, b):
        b = np.array(b, np.float64); lib.orc_eliminate_rhs(C.byref(self.op), np.ascontiguousarray(x), b); return b

    def pcg(self, b, dinv, rel_tol, abs_tol=0.0, max_iter=2000):
        x = np.zeros(self.n); hist = np.zeros(max_iter + 2); conv = C.c_int(0)
        dp = None if dinv is None else np.ascontiguousarray(dinv).ctypes.data_as(C.c_void_p)
        self._d = dinv
        its = lib.orc_pcg(C.byref(self.op), dp, np.ascontiguousarray(b), x, rel_tol, abs_tol, max_iter, hist, C.byref(conv))
        return x, its, hist[:its + 1], bool(conv.value)


def jacobi_setup(diag, ess):
    dinv = np.zeros_like(diag)
    lib.orc_jacobi_setup(len(diag), np.ascontiguousarray(diag), len(ess), np.ascontiguousarray(ess, np.int32), dinv)
    return dinv
