"""-m gpu parity tests: the HIP path, called through the C ABI (include/mfem_amd.h), against
(a) arrays produced by the reference itself (tests/golden) and (b) the CPU oracle on the
same inputs.  Integer/index work and the E->L sum: bit-exact.  Floating point: 1e-12
relative (north_star), tolerance written at each assert."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from conftest import golden_cases, load_case, splitmix  # noqa: E402

CASES = golden_cases()
TOL = 1e-12


EXTRA_SHAPES = [(1, 4, 6), (1, 5, 8), (0, 2, 4), (0, 3, 6), (0, 4, 6), (0, 5, 8)]   # the reference's "others"


def registered(kind, D, Q):
    """MFEM's 3D specialisation lists (bilininteg_diffusion_kernels.cpp:44-63, bilininteg_mass_kernels.cpp:46-68),
    plus p = 8 -- and (9,9) for diffusion -- which the reference leaves to its fallback"""
    return (Q in (D, D + 1) and 2 <= D <= 9) or (kind, D, Q) in EXTRA_SHAPES or (kind == 0 and (D, Q) in ((4, 6), (5, 8)))


def relerr(a, b):
    return np.max(np.abs(a - b))/max(np.max(np.abs(b)), 1e-300)


@pytest.fixture(scope="module")
def gh():
    import gpu_helpers
    from mfem_amd import capi
    capi.call("mfem_amd_init", 0)
    return gpu_helpers


@pytest.mark.parametrize("name", CASES)
def test_restriction(gh, name):
    from mfem_amd import capi
    c = load_case(name)
    nd = c["D1D"]**3
    x, gm = gh.dev(c["x"]), gh.dev(c["gather_map"], np.int32)
    xE = gh.zeros(c["NE"]*nd)
    capi.call("mfem_amd_restrict_gather", c["NE"]*nd, gh.ptr(gm), gh.ptr(x), gh.ptr(xE), None)
    assert np.array_equal(gh.host(xE), c["xE"])                       # bit-exact
    # E->L: same summation order as the reference -> bit-exact on the reference's own yE
    off, ind, yE = gh.dev(c["offsets"], np.int32), gh.dev(c["indices"], np.int32), gh.dev(c["yE"])
    y = gh.zeros(c["ndofs"])
    capi.call("mfem_amd_restrict_scatter", c["ndofs"], gh.ptr(off), gh.ptr(ind), gh.ptr(yE), gh.ptr(y), 0, 0, None)
    assert np.array_equal(gh.host(y), c["y"])                         # bit-exact
    capi.call("mfem_amd_restrict_scatter", c["ndofs"], gh.ptr(off), gh.ptr(ind), gh.ptr(yE), gh.ptr(y), 1, 0, None)
    assert np.array_equal(gh.host(y), c["y"] + c["y"])
    dE = gh.dev(c["diagE"])
    capi.call("mfem_amd_restrict_scatter", c["ndofs"], gh.ptr(off), gh.ptr(ind), gh.ptr(dE), gh.ptr(y), 0, 1, None)
    assert np.array_equal(gh.host(y), c["diag"])                      # AbsMultTranspose, bit-exact
    # mesh nodes (vdim = 3)
    ndn = c["nodes_D1D"]**3
    nodes, ngm = gh.dev(c["nodes"]), gh.dev(c["nodes_gather_map"], np.int32)
    nE = gh.zeros(c["NE"]*ndn*3)
    capi.call("mfem_amd_restrict_gather_vdim", c["NE"]*ndn, ndn, 3, c["nodes_byvdim"], c["nodes_ndofs"],
              gh.ptr(ngm), gh.ptr(nodes), gh.ptr(nE), None)
    assert np.array_equal(gh.host(nE), c["nodes_E"])


@pytest.mark.parametrize("name", CASES)
def test_geometry_and_setup(gh, name):
    from mfem_amd import capi
    c = load_case(name)
    Q, NE, Dn = c["Q1D"], c["NE"], c["nodes_D1D"]
    nE, W, coeff = gh.dev(c["nodes_E"]), gh.dev(c["W"]), gh.dev(c["coeff"])
    const_c = int(len(c["coeff"]) == 1)
    J, detJ = gh.zeros(9*Q**3*NE), gh.zeros(Q**3*NE)
    capi.call("mfem_amd_geom_jacobians_3d", Dn, Q, NE, gh.hB(c["nodes_B"]), gh.hB(c["nodes_G"]), gh.ptr(nE), gh.ptr(J), None)
    capi.call("mfem_amd_geom_det_3d", Q, NE, gh.ptr(J), gh.ptr(detJ), None)
    if "J" in c:
        assert relerr(gh.host(J), c["J"]) < 1e-13
    assert relerr(gh.host(detJ), c["detJ"]) < TOL
    ncomp = 6 if c["kind"] else 1
    D1, D2, det2 = gh.zeros(ncomp*Q**3*NE), gh.zeros(ncomp*Q**3*NE), gh.zeros(Q**3*NE)
    if c["kind"]:
        capi.call("mfem_amd_diffusion_setup_3d", Q, 1, NE, gh.ptr(W), gh.ptr(J), gh.ptr(coeff), const_c, gh.ptr(D1), None)
    else:
        capi.call("mfem_amd_mass_setup_3d", Q, NE, gh.ptr(W), gh.ptr(detJ), gh.ptr(coeff), const_c, gh.ptr(D1), None)
    assert relerr(gh.host(D1), c["pa_data"]) < TOL
    capi.call("mfem_amd_pa_setup_from_nodes_3d", c["kind"], Dn, Q, 1, NE, gh.hB(c["nodes_B"]), gh.hB(c["nodes_G"]),
              gh.ptr(W), gh.ptr(nE), gh.ptr(coeff), const_c, 0, gh.ptr(D2), gh.ptr(det2), None)
    assert relerr(gh.host(D2), c["pa_data"]) < TOL
    assert relerr(gh.host(det2), c["detJ"]) < TOL
    # blocked (HBM) layout: a pure permutation of the same values -> bit-identical to converting D2
    nblk = capi.lib.mfem_amd_pa_data_size(c["kind"], 1, Q, NE)
    assert nblk >= ncomp*Q**3*NE and capi.lib.mfem_amd_pa_data_size(c["kind"], 0, Q, NE) == ncomp*Q**3*NE
    D3, D4 = gh.zeros(nblk), gh.zeros(nblk)
    capi.call("mfem_amd_pa_setup_from_nodes_3d", c["kind"], Dn, Q, 1, NE, gh.hB(c["nodes_B"]), gh.hB(c["nodes_G"]),
              gh.ptr(W), gh.ptr(nE), gh.ptr(coeff), const_c, 1, gh.ptr(D3), None, None)
    capi.call("mfem_amd_pa_data_to_blocked", c["kind"], Q, NE, gh.ptr(D2), gh.ptr(D4), None)
    assert np.array_equal(gh.host(D3), gh.host(D4))
    assert np.array_equal(np.sort(gh.host(D4)[gh.host(D4) != 0]), np.sort(gh.host(D2)[gh.host(D2) != 0]))


@pytest.mark.parametrize("name", CASES)
def test_apply_and_diag(gh, name):
    from mfem_amd import capi
    import oracle_lib as orc
    c = load_case(name)
    D1, Q1, NE, nd = c["D1D"], c["Q1D"], c["NE"], c["D1D"]**3
    B, G = gh.hB(c["B"]), gh.hB(c["G"])
    pa, xE = gh.dev(c["pa_data"]), gh.dev(c["xE"])
    # AddMultPA semantics: yE += A xE, starting from a non-zero yE
    y0 = splitmix(7, NE*nd)
    yE = gh.dev(y0)
    if c["kind"]:
        capi.call("mfem_amd_pa_diffusion_apply_3d", D1, Q1, NE, 1, B, G, gh.ptr(pa), gh.ptr(xE), gh.ptr(yE), None)
    else:
        capi.call("mfem_amd_pa_mass_apply_3d", D1, Q1, NE, B, gh.ptr(pa), gh.ptr(xE), gh.ptr(yE), None)
    got = gh.host(yE) - y0
    scale = np.max(np.abs(c["yE"]))
    assert np.max(np.abs(got - c["yE"]))/scale < 1e-12 + 4*np.finfo(float).eps*np.max(np.abs(y0))/scale
    # fused L->E + deterministic E->L against the reference's L->L action
    op = gh.DeviceOperator(c)
    y = gh.host(op.apply_L2L(gh.dev(c["x"])))
    assert relerr(y, c["y"]) < TOL                                     # vs the reference
    assert relerr(gh.host(op.yE), c["yE"]) < TOL
    oop = orc.Operator(c["kind"], D1, Q1, NE, c["ndofs"], c["B"], c["G"], c["pa_data"], c["gather_map"],
                       c["offsets"], c["indices"])
    assert relerr(y, oop.mult(c["x"])) < TOL                           # vs the CPU oracle
    if capi.lib.mfem_amd_has_specialization(c["kind"], D1, Q1):
        # blocked pa_data layout: same arithmetic on permuted storage -> bit-identical
        opb = gh.DeviceOperator(c, layout=1)
        assert np.array_equal(gh.host(opb.apply_L2L(gh.dev(c["x"]))), y)
    # element and L diagonals
    for lay in (0, 1):
        dE = gh.zeros(NE*nd)
        pad = pa
        if lay:
            pad = gh.zeros(capi.lib.mfem_amd_pa_data_size(c["kind"], 1, Q1, NE))
            capi.call("mfem_amd_pa_data_to_blocked", c["kind"], Q1, NE, gh.ptr(pa), gh.ptr(pad), None)
        if c["kind"]:
            capi.call("mfem_amd_pa_diffusion_diag_3d", D1, Q1, NE, 1, B, G, gh.ptr(pad), lay, gh.ptr(dE), None)
        else:
            capi.call("mfem_amd_pa_mass_diag_3d", D1, Q1, NE, B, gh.ptr(pad), lay, gh.ptr(dE), None)
        assert relerr(gh.host(dE), c["diagE"]) < TOL
    assert capi.lib.mfem_amd_has_specialization(c["kind"], D1, Q1) == int(registered(c["kind"], D1, Q1))


@pytest.mark.parametrize("name", CASES)
def test_constrained_operator_and_pcg(gh, name):
    from mfem_amd import capi
    c = load_case(name)
    from mfem_amd import capi as _capi
    n, ess = c["ndofs"], c["ess_tdof"]
    layout = int(_capi.lib.mfem_amd_has_specialization(c["kind"], c["D1D"], c["Q1D"]) and c["NE"] % 2 == 0)
    op = gh.DeviceOperator(c, ess, layout=layout)     # alternate layouts over the cases
    y = gh.host(op.constrained_mult(gh.dev(c["x"])))
    assert relerr(y, c["y_constrained"]) < TOL
    assert np.array_equal(y[ess], c["x"][ess])
    # Jacobi from the L-diagonal
    diag, dinv = gh.dev(c["diag"]), gh.zeros(n)
    capi.call("mfem_amd_jacobi_setup", n, gh.ptr(diag), 1.0, len(ess), gh.ptr(op.ess), gh.ptr(dinv), None)
    z = gh.zeros(n)
    capi.call("mfem_amd_jacobi_apply", n, gh.ptr(dinv), gh.ptr(gh.dev(c["x"])), gh.ptr(z), None)
    assert relerr(gh.host(z), c["jacobi_x"]) < TOL
    # PCG: identical iteration count, solution to 1e-9 (CG amplifies rounding by cond(A)),
    # (B r, r) history to the 6 digits the reference prints
    rel_tol = 1e-10 if c["kind"] == 0 else 1e-6
    b = gh.dev(c["B_rhs"])
    for check_every in (1, 5):
        X, res, hist = op.pcg(b, dinv, rel_tol, check_every=check_every)
        assert res.iterations == c["cg_iters"] and bool(res.converged) == bool(c["cg_converged"])
        ref = c["cg_hist"]
        assert len(hist) == len(ref)
        # the reference log has 6 digits; the well-resolved part of the history must agree to
        # that, the tail (residual reduced by > 1e6) carries the dot-product-order rounding of
        # CG amplified by cond(A) and is held to 1e-3 relative
        big = ref > 1e-6*ref[0]
        assert np.max(np.abs(hist[big] - ref[big])/ref[big]) < 1e-5
        tail = (~big) & (ref > 1e-20*ref[0])
        if tail.any():
            assert np.max(np.abs(hist[tail] - ref[tail])/ref[tail]) < 1e-3
        assert relerr(gh.host(X), c["X"]) < 1e-9
    # bit-reproducible across runs and across check_every (no-op kernels after convergence)
    X1, _, h1 = op.pcg(b, dinv, rel_tol, check_every=1)
    X2, _, h2 = op.pcg(b, dinv, rel_tol, check_every=7)
    assert np.array_equal(gh.host(X1), gh.host(X2)) and np.array_equal(h1, h2)


@pytest.mark.parametrize("name", CASES[:4])
def test_pcg_streaming_policy_is_bit_identical(gh, name, monkeypatch):
    """Large solves load and store x, r and dinv non-temporally in the solver's vector kernels (vector_cg.hip: pcg_nt,
    on by default above 5.6e6 unknowns); MFEM_AMD_PCG_NT forces the choice.  A cache policy: same bits, counts, history."""
    from mfem_amd import capi
    c = load_case(name)
    n, ess = c["ndofs"], c["ess_tdof"]
    op = gh.DeviceOperator(c, ess)
    diag, dinv = gh.dev(c["diag"]), gh.zeros(n)
    capi.call("mfem_amd_jacobi_setup", n, gh.ptr(diag), 1.0, len(ess), gh.ptr(op.ess), gh.ptr(dinv), None)
    rel_tol = 1e-10 if c["kind"] == 0 else 1e-6
    b = gh.dev(c["B_rhs"])
    out = {}
    for nt in ("0", "1"):
        monkeypatch.setenv("MFEM_AMD_PCG_NT", nt)
        for jac in (dinv, None):
            X, r, h = op.pcg(b, jac, rel_tol, max_iter=300, check_every=4)
            out[nt, jac is None] = (gh.host(X), r.iterations, r.converged, r.final_nom, h)
    for key in (False, True):
        a, bb = out["0", key], out["1", key]
        assert np.array_equal(a[0], bb[0]) and a[1:4] == bb[1:4] and np.array_equal(a[4], bb[4])
    assert out["1", False][1] == c["cg_iters"]


def test_pcg_curvature_breakdown_counts(gh):
    """(A d, d) = 0: CGSolver::Mult leaves with converged = false (linalg/solvers.cpp:930-940 at iteration 0,
    :1011-1025 inside the loop, where the count has been advanced past the last residual it printed).
    Operator: zero coefficient, so the constrained operator is the identity on the essential dofs and 0 elsewhere."""
    c = dict(load_case(CASES[0]))
    c["pa_data"] = np.zeros_like(c["pa_data"])
    n = c["ndofs"]
    ess = np.arange(0, 4, dtype=np.int32)
    # (1) no essential dofs: A = 0, (A d, d) = 0 at the start -> 0 iterations, one history entry
    op0 = gh.DeviceOperator(c, None)
    b = np.ones(n)
    X, res, hist = op0.pcg(gh.dev(b), None, 1e-12, max_iter=10)
    assert (res.iterations, res.converged, res.history_count) == (0, 0, 1) and hist[0] == float(n)
    assert not gh.host(X).any()
    # (2) b = 1 on 4 essential and 4 other dofs: alpha = 2, r = (-1, 1), beta = 1, d = (0, 2): every step exact,
    #     (A d, d) = 0 at the second direction -> final_iter = 2 with the residuals of iterations 0 and 1 only
    op = gh.DeviceOperator(c, ess)
    b = np.zeros(n); b[:8] = 1.0
    for check_every in (1, 3):
        X, res, hist = op.pcg(gh.dev(b), None, 1e-12, max_iter=10, check_every=check_every)
        assert (res.iterations, res.converged, res.history_count) == (2, 0, 2), (res.iterations, res.converged, res.history_count)
        assert list(hist) == [8.0, 8.0] and res.final_nom == 8.0
        x = gh.host(X)
        assert np.array_equal(x[:8], 2.0*np.ones(8)) and not x[8:].any()


def test_vector_kernels(gh):
    from mfem_amd import capi
    for n in (1, 255, 4099, 1_000_003):
        a, b = splitmix(3, n) - 0.5, splitmix(4, n) - 0.25
        da, db, dz = gh.dev(a), gh.dev(b), gh.zeros(n)
        capi.call("mfem_amd_vec_axpby", n, 1.0, gh.ptr(da), -0.75, gh.ptr(db), gh.ptr(dz), None)
        assert np.array_equal(gh.host(dz), a + (-0.75)*b) or relerr(gh.host(dz), a - 0.75*b) < 1e-15
        ws = gh.zeros(capi.lib.mfem_amd_reduce_workspace_bytes(), torch.uint8)
        r1, r2 = gh.zeros(1), gh.zeros(1)
        capi.call("mfem_amd_vec_dot", n, gh.ptr(da), gh.ptr(db), gh.ptr(r1), gh.ptr(ws), None)
        capi.call("mfem_amd_vec_dot", n, gh.ptr(da), gh.ptr(db), gh.ptr(r2), gh.ptr(ws), None)
        exact = float(np.dot(a.astype(np.longdouble), b.astype(np.longdouble)))
        assert abs(gh.host(r1)[0] - exact) <= 1e-13*np.sum(np.abs(a*b))
        assert gh.host(r1)[0] == gh.host(r2)[0]                        # deterministic
        capi.call("mfem_amd_vec_fill", n, 2.5, gh.ptr(dz), None)
        assert np.all(gh.host(dz) == 2.5)
        # aliasing z = x allowed (linalg/vector.cpp:436-473)
        capi.call("mfem_amd_vec_axpby", n, 1.0, gh.ptr(da), 2.0, gh.ptr(db), gh.ptr(da), None)
        assert relerr(gh.host(da), a + 2.0*b) < 1e-15


def test_errors_are_reported(gh):
    from mfem_amd import capi
    x = gh.zeros(8)
    B = np.zeros(11*15)
    rc = capi.lib.mfem_amd_pa_mass_apply_3d(11, 15, 1, B, gh.ptr(x), gh.ptr(x), gh.ptr(x), None)
    assert rc != 0 and b"unsupported" in capi.lib.mfem_amd_last_error()
    # the non-symmetric (9-plane) D has no blocked layout: refused before anything is launched
    rc = capi.lib.mfem_amd_pa_diffusion_diag_3d(3, 4, 1, 0, B, B, gh.ptr(x), 1, gh.ptr(x), None)
    assert rc != 0 and b"symmetric" in capi.lib.mfem_amd_last_error()
    # empty inputs are fine
    capi.call("mfem_amd_restrict_gather", 0, None, None, None, None)
    capi.call("mfem_amd_pa_mass_apply_3d", 3, 4, 0, np.zeros(12), gh.ptr(x), gh.ptr(x), gh.ptr(x), None)


def test_trace_ranges_load_and_do_not_change_results(gh):
    """MFEM_AMD_TRACE=1: the stages of the path are wrapped in roctx ranges (csrc/common.h TraceRange, the role of
    MFEM_PERF_SCOPE); the roctx library is found at run time and the solve is the same solve."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    code = (
        "import sys; sys.path.insert(0, 'tests')\n"
        "import numpy as np, gpu_helpers as gh\n"
        "from conftest import golden_cases, load_case\n"
        "from mfem_amd import capi\n"
        "capi.call('mfem_amd_init', 0)\n"
        "c = load_case(golden_cases()[0])\n"
        "op = gh.DeviceOperator(c, c['ess_tdof'])\n"
        "n = c['ndofs']; diag, dinv = gh.dev(c['diag']), gh.zeros(n)\n"
        "capi.call('mfem_amd_jacobi_setup', n, gh.ptr(diag), 1.0, len(c['ess_tdof']), gh.ptr(op.ess), gh.ptr(dinv), None)\n"
        "X, res, hist = op.pcg(gh.dev(c['B_rhs']), dinv, 1e-10 if c['kind'] == 0 else 1e-6)\n"
        "assert res.iterations == c['cg_iters'], (res.iterations, c['cg_iters'])\n"
        "print('traced solve ok', res.iterations)\n")
    env = dict(os.environ, MFEM_AMD_TRACE="1")
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "traced solve ok" in r.stdout
    assert "ranges are off" not in r.stderr, r.stderr


@pytest.mark.parametrize("kind,D,Q", EXTRA_SHAPES)
def test_extra_registered_shapes(gh, kind, D, Q):
    """Over-integrated rules (Q1D = D1D + 2, + 3) that MFEM registers beside Q = P+1, P+2: the specialised kernel in
    both pa_data layouts against the CPU oracle (1e-12), the layouts against each other (bit-identical), the
    vector-space kernel against the scalar one per component (bit-identical), the diagonal against the oracle."""
    import oracle_lib as orc
    from mfem_amd import capi
    from mfem_amd.cartesian import CartesianProblem
    assert capi.lib.mfem_amd_has_specialization(kind, D, Q) == 1
    prob = CartesianProblem(D - 1, 3, 2, 2)
    ne, ndofs, nd = prob.ne, prob.ndofs, D**3
    B, G = orc.basis_1d(D - 1, orc.gauss_legendre(Q)[0])
    nc = 6 if kind else 1
    pa = 0.5 + splitmix(11, nc*Q**3*ne)
    x = splitmix(12, ndofs) - 0.5
    c = dict(kind=kind, D1D=D, Q1D=Q, NE=ne, ndofs=ndofs, B=B, G=G, pa_data=pa, gather_map=prob.gather_map,
             offsets=prob.offsets, indices=prob.indices)
    oop = orc.Operator(kind, D, Q, ne, ndofs, B, G, pa, prob.gather_map, prob.offsets, prob.indices)
    want = oop.mult(x)
    y0 = gh.host(gh.DeviceOperator(c, layout=0).apply_L2L(gh.dev(x)))
    op1 = gh.DeviceOperator(c, layout=1)
    y1 = gh.host(op1.apply_L2L(gh.dev(x)))
    assert relerr(y0, want) < TOL
    assert np.array_equal(y0, y1)
    # E-vector seam (AddMultPA: yE += A xE)
    xE, yE = gh.zeros(ne*nd), gh.zeros(ne*nd)
    gm = gh.dev(prob.gather_map, np.int32)
    capi.call("mfem_amd_restrict_gather", ne*nd, gh.ptr(gm), gh.ptr(gh.dev(x)), gh.ptr(xE), None)
    if kind:
        capi.call("mfem_amd_pa_diffusion_apply_3d", D, Q, ne, 1, gh.hB(B), gh.hB(G), gh.ptr(gh.dev(pa)), gh.ptr(xE), gh.ptr(yE), None)
    else:
        capi.call("mfem_amd_pa_mass_apply_3d", D, Q, ne, gh.hB(B), gh.ptr(gh.dev(pa)), gh.ptr(xE), gh.ptr(yE), None)
    # diagonal
    dE, dL = gh.zeros(ne*nd), gh.zeros(ndofs)
    if kind:
        capi.call("mfem_amd_pa_diffusion_diag_3d", D, Q, ne, 1, gh.hB(B), gh.hB(G), gh.ptr(gh.dev(pa)), 0, gh.ptr(dE), None)
    else:
        capi.call("mfem_amd_pa_mass_diag_3d", D, Q, ne, gh.hB(B), gh.ptr(gh.dev(pa)), 0, gh.ptr(dE), None)
    off, ind = gh.dev(prob.offsets, np.int32), gh.dev(prob.indices, np.int32)
    capi.call("mfem_amd_restrict_scatter", ndofs, gh.ptr(off), gh.ptr(ind), gh.ptr(dE), gh.ptr(dL), 0, 1, None)
    assert relerr(gh.host(dL), oop.diag()) < TOL
    # the E-vector of the seam equals the fused kernel's (same arithmetic): compare through E->L
    yL = gh.zeros(ndofs)
    capi.call("mfem_amd_restrict_scatter", ndofs, gh.ptr(off), gh.ptr(ind), gh.ptr(yE), gh.ptr(yL), 0, 0, None)
    assert relerr(gh.host(yL), want) < TOL
    # vector space, two components byNODES: component c of the vector apply = the scalar apply of component c
    vd = 2
    xv = splitmix(13, vd*ndofs) - 0.5
    yEv = gh.zeros(ne*nd*vd)
    for lay, pad in ((0, gh.dev(pa)), (1, op1.pa)):
        capi.call("mfem_amd_pa_apply_L2E_vdim", kind, lay, D, Q, ne, vd, 0, ndofs, gh.hB(B), gh.hB(G), gh.ptr(pad), 0, gh.ptr(gm), None,
                  gh.ptr(gh.dev(xv)), gh.ptr(yEv), None)
        got = gh.host(yEv).reshape(ne, vd, nd)
        for cmp_ in range(vd):
            sc = gh.DeviceOperator(c, layout=lay)
            sc.apply_L2L(gh.dev(xv[cmp_*ndofs:(cmp_ + 1)*ndofs]))
            assert np.array_equal(got[:, cmp_, :].reshape(-1), gh.host(sc.yE))
