"""-m gpu: mfem_amd_restrict_scatter_subrows (the chunk-pipelined E->L of include/mfem_amd.h) against
mfem_amd_restrict_scatter on the reference's own tables and E-vectors: bit-identical for every chunk size, with the
chunks' sub-rows submitted in element order (the additions are the reference's, fem/restriction.cpp:152-180)."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from conftest import golden_cases, load_case  # noqa: E402

sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "tools"))

CASES = [c for c in golden_cases() if c.startswith("cart3_bp3_p")][:4] + [c for c in golden_cases() if "toroid" in c][:1]


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("chunk_elems", [1, 2, 5, 1000])
def test_subrows_bit_identical(name, chunk_elems):
    import gpu_helpers as gh
    from chunk_probe import chunk_tables
    from mfem_amd import capi
    capi.call("mfem_amd_init", 0)
    c = load_case(name)
    D3, ndofs = c["D1D"]**3, c["ndofs"]
    off, ind, yE = c["offsets"], c["indices"], c["yE"]
    code, sub_off, rel, bounds = chunk_tables(off, ind, D3, chunk_elems)
    d_off, d_ind, d_yE = gh.dev(off, np.int32), gh.dev(ind, np.int32), gh.dev(yE)
    y_ref = gh.zeros(ndofs)
    capi.call("mfem_amd_restrict_scatter", ndofs, gh.ptr(d_off), gh.ptr(d_ind), gh.ptr(d_yE), gh.ptr(y_ref), 0, 0, None)
    d_code, d_soff, d_rel = gh.dev(code, np.int32), gh.dev(sub_off, np.int32), gh.dev(rel, np.int32)
    y = torch.full((ndofs,), float("nan"), dtype=torch.float64, device=gh.DEV)     # every row's first sub-row starts from 0
    p = lambda t, byte_off=0: C.c_void_p(t.data_ptr() + byte_off)
    nchunks = len(bounds) - 1
    assert nchunks == -(-c["NE"]//chunk_elems)
    for k in range(nchunks):                       # element order; each chunk's part of the E-vector addressed from its start
        s0, s1 = int(bounds[k]), int(bounds[k + 1])
        if s1 == s0:
            continue
        capi.call("mfem_amd_restrict_scatter_subrows", s1 - s0, p(d_code, 4*s0), p(d_soff, 4*s0), gh.ptr(d_rel),
                  p(d_yE, 8*k*chunk_elems*D3), gh.ptr(y), None)
    assert np.array_equal(gh.host(y), gh.host(y_ref))                                # bit-exact
    assert np.array_equal(gh.host(y), c["y"])
    capi.call("mfem_amd_restrict_scatter_subrows", 0, None, None, None, None, None, None)   # empty input
