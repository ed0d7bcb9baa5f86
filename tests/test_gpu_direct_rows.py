"""-m gpu: element-interior rows written by the element kernel and the packed E-vector that goes with them
(include/mfem_amd.h: mfem_amd_interior_rows, mfem_amd_pack_indices, mfem_amd_pa_apply_L2L, mfem_amd_pa_operator.direct_rows).  The values are the ones the E-vector would have carried
and the rows have one entry each, so everything here is bit-exact against the plain two-kernel form; the PCG's dot
product sums in another order and is held to 1e-12 relative."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from conftest import golden_cases, load_case  # noqa: E402

CASES = golden_cases()


@pytest.fixture(scope="module")
def gh():
    import gpu_helpers
    from mfem_amd import capi
    capi.call("mfem_amd_init", 0)
    return gpu_helpers


def interior_rows(gh, D, ne, ndofs, gm, off):
    from mfem_amd import capi
    work = gh.zeros(1, torch.int32)
    dr = C.c_int64(-1)
    capi.call("mfem_amd_interior_rows", D, ne, ndofs, gh.ptr(gm), gh.ptr(off), gh.ptr(work), C.byref(dr), None)
    return int(dr.value)


@pytest.mark.parametrize("name", CASES)
def test_golden_meshes_direct_rows_bit_identical(gh, name, monkeypatch):
    """The reference's own numbering (the golden cases were dumped from it): the probe finds the interior rows of every
    case with D1D >= 3, and the constrained operator with them written directly equals the plain form, both layouts."""
    from mfem_amd import capi
    c = load_case(name)
    D, NE, n = c["D1D"], c["NE"], c["ndofs"]
    monkeypatch.setenv("MFEM_AMD_DIRECT", "2")       # every shape, not only where the default policy uses it (D1D >= 5)
    for layout in (0, 1):
        if layout and not (capi.lib.mfem_amd_has_specialization(c["kind"], D, c["Q1D"]) and NE % 2 == 0):
            continue
        op = gh.DeviceOperator(c, c["ess_tdof"], layout=layout)
        dr = interior_rows(gh, D, NE, n, op.gm, op.off)
        assert dr == (NE*(D - 2)**3 if D >= 3 else 0), (name, dr)
        x = gh.dev(c["x"])
        y0 = gh.host(op.constrained_mult(x))
        if not dr:
            continue
        # direct rows + the packed E-vector (mfem_amd_pack_indices): same values, same order of additions, same bits
        pind = gh.zeros(op.ind.numel(), torch.int32)
        capi.call("mfem_amd_pack_indices", D, op.ind.numel(), gh.ptr(op.ind), gh.ptr(pind), None)
        ns = D**3 - (D - 2)**3
        live = gh.host(pind)[:int(c["offsets"][n - dr])]
        assert live.min() >= 0 and live.max() < NE*ns and len(np.unique(live)) == len(live) == NE*ns
        op.op.direct_rows = dr
        assert np.array_equal(gh.host(op.constrained_mult(x)), y0)       # no table: the plain form still
        op.op.packed_indices = pind.data_ptr()
        # what lies beyond the NE*NS packed entries is never touched, and nothing stale is read: poison all of it
        op.yE.fill_(float("nan"))
        assert np.array_equal(gh.host(op.constrained_mult(x)), y0)
        assert torch.isnan(op.yE[NE*ns:]).all() and not torch.isnan(op.yE[:NE*ns]).any()
        # an element-interior dof in the essential list (a pinned dof): its row takes x like any other
        ess2 = np.unique(np.concatenate([np.asarray(c["ess_tdof"], np.int64), [n - 1, n - dr, n - dr//2 - 1]])).astype(np.int32)
        op2 = gh.DeviceOperator(c, ess2, layout=layout)
        yref = gh.host(op2.constrained_mult(x))
        assert np.array_equal(yref[ess2], c["x"][ess2])
        op2.op.direct_rows = dr
        op2.op.packed_indices = pind.data_ptr()
        assert np.array_equal(gh.host(op2.constrained_mult(x)), yref)
        # ... and inside the solver (the fused (d, A d) takes its direct-row part from the patched value)
        dinv = gh.dev(1.0/np.where(np.abs(c["diag"]) > 0, c["diag"], 1.0)); dinv[gh.dev(ess2, np.int64)] = 1.0
        b = gh.dev(np.where(np.isin(np.arange(n), ess2), 0.0, c["B_rhs"]))
        op3 = gh.DeviceOperator(c, ess2, layout=layout)
        X0, r0, h0 = op3.pcg(b, dinv, 1e-8, max_iter=300)
        X1, r1, h1 = op2.pcg(b, dinv, 1e-8, max_iter=300)
        assert r1.iterations == r0.iterations and r1.converged == r0.converged
        assert np.max(np.abs(h1 - h0)) <= 1e-11*abs(h0[0])
        assert torch.max(torch.abs(X1 - X0)).item() <= 1e-9*max(1e-300, torch.max(torch.abs(X0)).item())
    # a gather map that breaks the property anywhere is refused
    if D >= 3:
        gm = np.array(c["gather_map"], np.int32).reshape(NE, D, D, D)
        if NE > 1 and D > 3:
            gm[0, 1, 1, 1], gm[0, 1, 1, 2] = gm[0, 1, 1, 2], gm[0, 1, 1, 1]
            assert interior_rows(gh, D, NE, n, gh.dev(gm.reshape(-1), np.int32), op.off) == 0


@pytest.mark.parametrize("problem,p,n", [("bp1", 2, 5), ("bp3", 2, 4), ("bp3", 3, 4), ("bp1", 4, 3), ("bp3", 4, 4), ("bp5", 4, 3),
                                         ("bp3", 5, 3), ("bp1", 6, 2), ("bp3", 6, 3), ("bp5", 6, 2), ("bp3", 7, 2), ("bp1", 8, 2),
                                         ("bp3", 8, 2), ("bp5", 8, 2)])
def test_interior_last_numbering_matches_lexicographic(gh, problem, p, n, monkeypatch):
    from mfem_amd.bp import BakeoffProblem
    dev = torch.device("cuda", 0)
    lex = BakeoffProblem(problem, p, (n, n, n + 1), dev, device_tables=True)
    il = BakeoffProblem(problem, p, (n, n, n + 1), dev, device_tables=True, numbering="interior-last")
    assert lex.direct == 0 and il.direct == il.ne*(p - 1)**3 and il.op.direct_rows == il.direct
    monkeypatch.setenv("MFEM_AMD_DIRECT_ROWS", "0")
    il0 = BakeoffProblem(problem, p, (n, n, n + 1), dev, device_tables=True, numbering="interior-last")
    monkeypatch.delenv("MFEM_AMD_DIRECT_ROWS")
    assert il0.direct == 0
    g = torch.Generator(device="cpu"); g.manual_seed(7)
    xl = torch.rand(lex.ndofs, dtype=torch.float64, generator=g).to(dev)
    xi = torch.empty_like(xl); xi[il.perm] = xl
    yl, yi, y0 = torch.empty_like(xl), torch.full_like(xl, float("nan")), torch.empty_like(xl)
    monkeypatch.setenv("MFEM_AMD_DIRECT", "2")       # every order, not only where the default policy uses it (p >= 4)
    lex.mult(xl, yl); il.mult(xi, yi); il0.mult(xi, y0)
    torch.cuda.synchronize()
    assert torch.equal(yi, y0)                       # direct rows vs the plain form, same numbering: same bits
    assert torch.equal(yi[il.perm], yl)              # and the same bits as the lexicographic numbering, row by row
    # constrained operator + Jacobi-PCG: same counts; the dot products sum in a different order
    b = torch.rand(lex.ndofs, dtype=torch.float64, generator=g).to(dev)
    bi = torch.empty_like(b); bi[il.perm] = b
    bi[il.ess.long()] = 0.0
    # (the default policy writes the interior rows directly wherever the numbering allows, inside the solver too, and takes
    # their part of (d, A d) in the transpose-gather kernel -- vector_cg.hip: constrained_mult; MFEM_AMD_DIRECT=0 never)
    monkeypatch.delenv("MFEM_AMD_DIRECT")
    yd = torch.full_like(xl, float("nan"))
    il.mult(xi, yd)                                  # default policy
    torch.cuda.synchronize()
    assert torch.equal(yd, y0)
    X3, r3, h3 = il.pcg(bi, 1e-8, max_iter=400, check_every=3, want_hist=True)
    monkeypatch.setenv("MFEM_AMD_DIRECT", "2")
    X1, r1, h1 = il.pcg(bi, 1e-8, max_iter=400, check_every=3, want_hist=True)
    X0, r0, h0 = il0.pcg(bi, 1e-8, max_iter=400, check_every=3, want_hist=True)
    assert torch.equal(X3, X1) and np.array_equal(h3, h1)
    monkeypatch.setenv("MFEM_AMD_DIRECT", "0")
    X4, r4, h4 = il.pcg(bi, 1e-8, max_iter=400, check_every=3, want_hist=True)
    assert torch.equal(X4, X0) and np.array_equal(h4, h0)          # the plain form on the same numbering: same bits
    monkeypatch.setenv("MFEM_AMD_DIRECT", "2")
    assert (r1.iterations, r1.converged) == (r0.iterations, r0.converged) and r1.converged
    assert np.max(np.abs(h1 - h0)/h0[0]) < 1e-12
    assert torch.max(torch.abs(X1 - X0)).item() <= 1e-10*torch.max(torch.abs(X0)).item()
    # run-to-run reproducible
    X2, r2, h2 = il.pcg(bi, 1e-8, max_iter=400, check_every=7, want_hist=True)
    assert torch.equal(X1, X2) and np.array_equal(h1, h2)
    monkeypatch.setenv("MFEM_AMD_DIRECT", "0")
    il.mult(xi, yi)
    torch.cuda.synchronize()
    assert torch.equal(yi, y0)
    assert il.pind is not None
    # the two halves run separately (bench.py's kernel timings) give the same y too
    yk = torch.full_like(xl, float("nan"))
    il.l2e(xi, y=yk); il.e2l(yk)
    torch.cuda.synchronize()
    assert torch.equal(yk, y0)


def test_l2l_rejects_bad_direct_rows(gh):
    from mfem_amd import capi
    c = load_case(CASES[0])
    op = gh.DeviceOperator(c, None)
    x, y = gh.dev(c["x"]), gh.zeros(c["ndofs"])
    rc = capi.lib.mfem_amd_pa_apply_L2L(c["kind"], 0, c["D1D"], c["Q1D"], c["NE"], c["ndofs"], op.B, op.G, gh.ptr(op.pa),
                                        gh.ptr(op.gm), gh.ptr(op.off), gh.ptr(op.ind), gh.ptr(op.ind), 5, gh.ptr(x), gh.ptr(op.yE),
                                        gh.ptr(y), None)
    assert rc != 0 and b"direct_rows" in capi.lib.mfem_amd_last_error()
    if c["D1D"] >= 3:
        dr = c["NE"]*(c["D1D"] - 2)**3
        rc = capi.lib.mfem_amd_pa_apply_L2L(c["kind"], 0, c["D1D"], c["Q1D"], c["NE"], c["ndofs"], op.B, op.G, gh.ptr(op.pa),
                                            gh.ptr(op.gm), gh.ptr(op.off), gh.ptr(op.ind), None, dr, gh.ptr(x), gh.ptr(op.yE),
                                            gh.ptr(y), None)
        assert rc != 0 and b"packed_indices" in capi.lib.mfem_amd_last_error()
