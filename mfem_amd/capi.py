"""ctypes binding of include/mfem_amd.h.  Device pointers are plain integers
(``tensor.data_ptr()``); nothing here computes anything on the CPU."""
import ctypes as C
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
_SO = Path(os.environ.get("MFEM_AMD_LIB", _HERE / "libmfem_amd.so"))     # MFEM_AMD_LIB: a tuning build of the same ABI
if not _SO.exists():
    raise ImportError(
        f"{_SO} is missing: build the HIP extension first "
        "(python -c 'import __graft_entry__ as g; g.build()' or make -C mfem_amd/csrc). "
        "mfem_amd has no CPU fallback.")
# One HIP runtime per process: PyTorch ships its own libamdhip64 and the drivers here use torch
# for device memory, so let torch load its runtime first -- the library's libamdhip64.so.7
# dependency then resolves to the copy already in the process instead of a second runtime
# (two runtimes do not share the device: whichever initialises second sees no GPU).
try:
    import torch  # noqa: F401
except ImportError:
    pass
lib = C.CDLL(str(_SO))

i64, vp, dbl, cint = C.c_int64, C.c_void_p, C.c_double, C.c_int
_hd = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")   # HOST double table


class PaOperator(C.Structure):
    """struct mfem_amd_pa_operator"""
    _fields_ = [("kind", cint), ("pa_layout", cint), ("D1D", cint), ("Q1D", cint), ("ne", i64), ("ndofs", i64),
                ("B", vp), ("G", vp), ("pa_data", vp), ("gather_map", vp),
                ("masked_gather_map", vp), ("offsets", vp), ("indices", vp),
                ("ness", i64), ("ess", vp), ("ess_marker", vp), ("yE", vp),
                ("vdim", cint), ("byvdim", cint), ("pa_comp_stride", i64), ("direct_rows", i64),
                ("packed_indices", vp), ("e_split", i64), ("reverse_overlap", cint), ("ghost_rows", vp), ("n_ghost_rows", i64)]


AR_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p)
EX_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)


class Comm(C.Structure):
    """struct mfem_amd_comm (reverse_begin / reverse_end are optional: NULL when the transport does not split P^T)"""
    _fields_ = [("ctx", vp), ("allreduce_sum", AR_FN), ("forward", EX_FN), ("reverse", EX_FN),
                ("reverse_begin", EX_FN), ("reverse_end", EX_FN)]


class PcgResult(C.Structure):
    """struct mfem_amd_pcg_result"""
    _fields_ = [("iterations", cint), ("converged", cint), ("initial_nom", dbl), ("final_nom", dbl), ("history_count", cint)]


_SIGS = {
    "mfem_amd_init": (cint, [cint]),
    "mfem_amd_last_error": (C.c_char_p, []),
    "mfem_amd_report_kernels": (None, [cint]),
    "mfem_amd_has_specialization": (cint, [cint, cint, cint]),
    "mfem_amd_restrict_gather": (cint, [i64, vp, vp, vp, vp]),
    "mfem_amd_restrict_gather_vdim": (cint, [i64, cint, cint, cint, i64, vp, vp, vp, vp]),
    "mfem_amd_restrict_scatter": (cint, [i64, vp, vp, vp, vp, cint, cint, vp]),
    "mfem_amd_restrict_scatter_subrows": (cint, [i64, vp, vp, vp, vp, vp, vp]),
    "mfem_amd_geom_jacobians_3d": (cint, [cint, cint, i64, _hd, _hd, vp, vp, vp]),
    "mfem_amd_geom_det_3d": (cint, [cint, i64, vp, vp, vp]),
    "mfem_amd_diffusion_setup_3d": (cint, [cint, cint, i64, vp, vp, vp, cint, vp, vp]),
    "mfem_amd_mass_setup_3d": (cint, [cint, i64, vp, vp, vp, cint, vp, vp]),
    "mfem_amd_pa_setup_from_nodes_3d": (cint, [cint, cint, cint, cint, i64, _hd, _hd, vp, vp, vp, cint, cint, vp, vp, vp]),
    "mfem_amd_pa_data_size": (i64, [cint, cint, cint, i64]),
    "mfem_amd_pa_data_to_blocked": (cint, [cint, cint, i64, vp, vp, vp]),
    "mfem_amd_pa_data_from_blocked": (cint, [cint, cint, i64, vp, vp, vp]),
    "mfem_amd_domain_lf_3d": (cint, [cint, cint, i64, vp, vp, vp, vp]),
    "mfem_amd_pa_diffusion_apply_3d": (cint, [cint, cint, i64, cint, _hd, _hd, vp, vp, vp, vp]),
    "mfem_amd_pa_mass_apply_3d": (cint, [cint, cint, i64, _hd, vp, vp, vp, vp]),
    "mfem_amd_pa_diffusion_diag_3d": (cint, [cint, cint, i64, cint, _hd, _hd, vp, cint, vp, vp]),
    "mfem_amd_pa_mass_diag_3d": (cint, [cint, cint, i64, _hd, vp, cint, vp, vp]),
    "mfem_amd_pa_apply_L2E": (cint, [cint, cint, cint, cint, i64, _hd, _hd, vp, vp, vp, vp, vp]),
    "mfem_amd_evec_add_marked": (cint, [i64, cint, cint, vp, vp, vp, vp, vp]),
    "mfem_amd_evec_zero_unmarked": (cint, [i64, cint, cint, vp, vp, vp, vp]),
    "mfem_amd_restrict_scatter_vdim": (cint, [i64, cint, cint, cint, vp, vp, vp, vp, cint, cint, vp]),
    "mfem_amd_pa_apply_L2E_vdim": (cint, [cint, cint, cint, cint, i64, cint, cint, i64, _hd, _hd, vp, i64, vp, vp, vp, vp, vp]),
    "mfem_amd_pa_vector_diffusion_apply_3d": (cint, [cint, cint, i64, cint, cint, _hd, _hd, vp, vp, vp, vp]),
    "mfem_amd_pa_vector_mass_apply_3d": (cint, [cint, cint, i64, cint, cint, _hd, vp, vp, vp, vp]),
    "mfem_amd_vec_replicate_vdim": (cint, [i64, cint, cint, vp, vp, vp, vp]),
    "mfem_amd_set_marker": (cint, [i64, vp, vp, vp]),
    "mfem_amd_mask_gather_map": (cint, [i64, vp, vp, vp, vp]),
    "mfem_amd_set_subvector": (cint, [i64, vp, vp, dbl, vp, vp]),
    "mfem_amd_vec_axpby": (cint, [i64, dbl, vp, dbl, vp, vp, vp]),
    "mfem_amd_vec_fill": (cint, [i64, dbl, vp, vp]),
    "mfem_amd_reduce_workspace_bytes": (i64, []),
    "mfem_amd_vec_dot": (cint, [i64, vp, vp, vp, vp, vp]),
    "mfem_amd_vec_dot_noalloc": (cint, [i64, vp, vp, vp, vp]),
    "mfem_amd_jacobi_setup": (cint, [i64, vp, dbl, i64, vp, vp, vp]),
    "mfem_amd_jacobi_apply": (cint, [i64, vp, vp, vp, vp]),
    "mfem_amd_pa_constrained_mult": (cint, [C.POINTER(PaOperator), vp, vp, vp]),
    "mfem_amd_interior_rows": (cint, [cint, i64, i64, vp, vp, vp, C.POINTER(i64), vp]),
    "mfem_amd_pa_apply_L2L": (cint, [cint, cint, cint, cint, i64, i64, _hd, _hd, vp, vp, vp, vp, vp, i64, vp, vp, vp, vp]),
    "mfem_amd_pack_indices": (cint, [cint, i64, vp, vp, vp]),
    "mfem_amd_pa_apply_L2E_packed": (cint, [cint, cint, cint, cint, i64, _hd, _hd, vp, vp, vp, vp, vp, vp]),
    "mfem_amd_pcg_workspace_bytes": (i64, [i64, cint]),
    "mfem_amd_shared_pack": (cint, [i64, vp, vp, vp, vp]),
    "mfem_amd_shared_unpack": (cint, [i64, vp, vp, vp, vp]),
    "mfem_amd_shared_accumulate": (cint, [i64, vp, vp, vp, vp, vp, vp]),
    "mfem_amd_pa_par_mult": (cint, [C.POINTER(PaOperator), C.POINTER(Comm), vp, vp, vp]),
    "mfem_amd_pcg_pa_par": (cint, [C.POINTER(PaOperator), C.POINTER(Comm), vp, vp, vp, dbl, dbl, cint, cint, vp,
                                   C.POINTER(PcgResult), vp, vp]),
    "mfem_amd_pcg_pa": (cint, [C.POINTER(PaOperator), vp, vp, vp, dbl, dbl, cint, cint, vp,
                               C.POINTER(PcgResult), vp, vp]),
}
EXPORTS = sorted(_SIGS)
for _n, (_r, _a) in _SIGS.items():
    _f = getattr(lib, _n)          # AttributeError here = header/library mismatch: fail loudly
    _f.restype, _f.argtypes = _r, _a


class MfemAmdError(RuntimeError):
    pass


def check(rc):
    """MFEM's convention at this boundary is abort-with-message; in Python we raise."""
    if rc != 0:
        raise MfemAmdError(lib.mfem_amd_last_error().decode())


def call(name, *args):
    check(getattr(lib, name)(*args))
