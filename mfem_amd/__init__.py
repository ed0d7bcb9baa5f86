"""mfem_amd -- MI355X-native partial-assembly operator-apply + Jacobi-PCG path of MFEM.

The product is the C-ABI library ``libmfem_amd.so`` (hand-written HIP for gfx950, see
``include/mfem_amd.h``) and the C++ host mirror of MFEM's interface in ``mfem_amd/host``.
This Python package is only the thin ctypes loader used by the tests and ``bench.py``;
PyTorch is used for device memory, streams and ``torch.distributed`` -- plumbing only.
There is NO CPU fallback: importing :mod:`mfem_amd.capi` fails loudly if the HIP library
is missing.
"""
from . import capi  # noqa: F401
