"""-m gpu: vector H1 spaces (VectorMassIntegrator / VectorDiffusionIntegrator, CEED BP2/BP4/BP6)
through the C ABI.

The reference applies the scalar operator to every component
(fem/integ/bilininteg_vecdiffusion_pa.hpp:108-170: `for i < SDIM` around the scalar kernel body with
the component's (Q^3,6) block of pa_data; fem/restriction.cpp:152-180 for the vdim restriction).
The scalar path is pinned to the reference's own outputs (test_gpu_parity.py), so the checks here
are BIT-equality of every component with the scalar path on the golden cases, for both L-vector
orderings, both pa_data layouts, shared and per-component pa_data, the generic fallback kernel, the
constrained operator and the fused PCG.  tests/test_gpu_vector_ref.py holds the reference's own
vector outputs."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from conftest import load_case          # noqa: E402

CASES = ["cart2_bp3_p1", "cart3_bp3_p2", "cart2_bp3_p4", "cart2_bp1_p3", "cart2_bp5_p5", "cart2_bp3_p7",
         "toroid-hex_diff_p2", "fichera-q3_massvar_p2", "cart2_bp1_p8"]


def vdof(i, c, ndofs, vd, byvdim):
    return i*vd + c if byvdim else c*ndofs + i


def to_vec(comps, byvdim):
    """L-vector of a vector space from its components (list of equal-length arrays)."""
    a = np.stack(comps, axis=1 if byvdim else 0)
    return np.ascontiguousarray(a.reshape(-1))


def split(v, ndofs, vd, byvdim):
    a = np.asarray(v).reshape((ndofs, vd) if byvdim else (vd, ndofs))
    return [np.ascontiguousarray(a[:, c] if byvdim else a[c]) for c in range(vd)]


@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("byvdim", [0, 1])
@pytest.mark.parametrize("layout", [0, 1])
def test_vector_apply_is_the_scalar_apply_per_component(name, byvdim, layout):
    from gpu_helpers import DeviceOperator, dev, host, ptr, zeros
    from mfem_amd import capi
    capi.call("mfem_amd_init", 0)
    c = load_case(name)
    kind, D, Q, NE, nd = int(c["kind"]), int(c["D1D"]), int(c["Q1D"]), int(c["NE"]), int(c["ndofs"])
    if layout and not capi.lib.mfem_amd_has_specialization(kind, D, Q):
        pytest.skip("the blocked layout needs a specialised (D1D,Q1D)")
    vd = 3
    op = DeviceOperator(c, layout=layout)
    rng = np.random.default_rng(7)
    comps = [rng.random(nd) for _ in range(vd)]
    x = dev(to_vec(comps, byvdim))
    yE = zeros(NE*D**3*vd)
    capi.call("mfem_amd_pa_apply_L2E_vdim", kind, layout, D, Q, NE, vd, byvdim, nd, op.B, op.G, ptr(op.pa), 0,
              ptr(op.gm), None, ptr(x), ptr(yE), None)
    y = zeros(nd*vd)
    capi.call("mfem_amd_restrict_scatter_vdim", nd, D**3, vd, byvdim, ptr(op.off), ptr(op.ind), ptr(yE), ptr(y), 0, 0, None)
    yEv = host(yE).reshape(NE, vd, D**3)
    yv = split(host(y), nd, vd, byvdim)
    for k in range(vd):
        xk = dev(comps[k])
        ys = op.apply_L2L(xk)                                 # scalar path (pinned to the reference)
        assert np.array_equal(yEv[:, k, :].reshape(-1), host(op.yE)), (name, k)     # E-vector (D^3,vd,NE)
        assert np.array_equal(yv[k], host(ys)), (name, k)
    # add and abs variants of the vdim transpose-gather
    y2 = dev(to_vec([np.full(nd, 1.0 + k) for k in range(vd)], byvdim))
    capi.call("mfem_amd_restrict_scatter_vdim", nd, D**3, vd, byvdim, ptr(op.off), ptr(op.ind), ptr(yE), ptr(y2), 1, 0, None)
    for k, part in enumerate(split(host(y2), nd, vd, byvdim)):
        assert np.array_equal(part, (1.0 + k) + yv[k])
    ya = zeros(nd*vd)
    capi.call("mfem_amd_restrict_scatter_vdim", nd, D**3, vd, byvdim, ptr(op.off), ptr(op.ind), ptr(yE), ptr(ya), 0, 1, None)
    assert np.array_equal(host(ya), host(y))                  # H1: no signs, abs == plain


@pytest.mark.parametrize("name", ["cart2_bp3_p3", "cart2_bp1_p4", "toroid-hex_diff_p3", "fichera-q3_massvar_p2"])
def test_vector_integrator_seam_with_per_component_data(name):
    """VectorDiffusion/VectorMassIntegrator::AddMultPA on E-vectors with the reference's pa_data layout
    ((Q^3,9,vdim,NE), six planes used / (Q^3,coeff_vdim,NE)) and a different coefficient per component: every component
    equals the scalar integrator applied with that component's data, and yE is ADDED to."""
    from gpu_helpers import dev, hB, host, ptr, zeros
    from mfem_amd import capi
    capi.call("mfem_amd_init", 0)
    c = load_case(name)
    kind, D, Q, NE = int(c["kind"]), int(c["D1D"]), int(c["Q1D"]), int(c["NE"])
    vd, nc, Q3, D3 = 3, (6 if kind else 1), Q**3, D**3
    B, G = hB(c["B"]), hB(c["G"])
    pa = np.asarray(c["pa_data"]).reshape(NE, nc, Q3)
    scale = [1.0, 0.5, 2.0]                                   # exact in binary: comp k = scale[k] * scalar operator
    pav = np.stack([pa*s for s in scale], axis=1)            # (NE, vd, nc, Q3) = (Q^3,nc,vd,NE) column-major
    if kind:
        # the reference allocates dim*dim = 9 planes per block and fills the symmetric 6: NaN the other 3
        pav = np.ascontiguousarray(np.concatenate([pav, np.full((NE, vd, 3, Q3), np.nan)], axis=2))
    rng = np.random.default_rng(3)
    xE = rng.random((NE, vd, D3))
    y0 = rng.random((NE, vd, D3))
    yE, pav_d, xE_d, pa_d = dev(y0.copy()), dev(pav), dev(xE), dev(pa)      # named: the buffers must outlive the calls
    if kind:
        capi.call("mfem_amd_pa_vector_diffusion_apply_3d", D, Q, NE, vd, vd, B, G, ptr(pav_d), ptr(xE_d), ptr(yE), None)
    else:
        capi.call("mfem_amd_pa_vector_mass_apply_3d", D, Q, NE, vd, vd, B, ptr(pav_d), ptr(xE_d), ptr(yE), None)
    got = host(yE).reshape(NE, vd, D3)
    for k in range(vd):
        ys = dev(np.ascontiguousarray(y0[:, k, :]))
        xs = dev(np.ascontiguousarray(xE[:, k, :]))
        pk = dev(np.ascontiguousarray(pav[:, k, :nc]))
        if kind:
            capi.call("mfem_amd_pa_diffusion_apply_3d", D, Q, NE, 1, B, G, ptr(pk), ptr(xs), ptr(ys), None)
        else:
            capi.call("mfem_amd_pa_mass_apply_3d", D, Q, NE, B, ptr(pk), ptr(xs), ptr(ys), None)
        assert np.array_equal(got[:, k, :], host(ys).reshape(NE, D3)), (name, k)
    if not kind:                                              # scalar coefficient: one shared (Q^3,NE) array
        yE2 = zeros(NE*vd*D3)
        capi.call("mfem_amd_pa_vector_mass_apply_3d", D, Q, NE, vd, 1, B, ptr(pa_d), ptr(xE_d), ptr(yE2), None)
        ys, x1 = zeros(NE*D3), dev(np.ascontiguousarray(xE[:, 1, :]))
        capi.call("mfem_amd_pa_mass_apply_3d", D, Q, NE, B, ptr(pa_d), ptr(x1), ptr(ys), None)
        assert np.array_equal(host(yE2).reshape(NE, vd, D3)[:, 1, :], host(ys).reshape(NE, D3))
    with pytest.raises(capi.MfemAmdError, match="matrix coefficients"):
        capi.call("mfem_amd_pa_vector_mass_apply_3d", D, Q, NE, vd, vd*vd, B, ptr(pav_d), ptr(xE_d), ptr(yE), None)


@pytest.mark.parametrize("name", ["cart3_bp3_p2", "cart2_bp1_p3", "toroid-hex_diff_p2"])
@pytest.mark.parametrize("byvdim", [0, 1])
def test_constrained_vector_operator_and_pcg(name, byvdim):
    """ConstrainedOperator on a vector space (essential dofs on every component), and the fused PCG:
    the vector system is block diagonal, so with a per-component Jacobi the solve of (b, b/2, 2b) stops
    at the scalar solve's iteration and returns (X, X/2, 2X)."""
    from gpu_helpers import DeviceOperator, dev, host, ptr, zeros
    from mfem_amd import capi
    capi.call("mfem_amd_init", 0)
    c = load_case(name)
    kind, D, Q, NE, nd = int(c["kind"]), int(c["D1D"]), int(c["Q1D"]), int(c["NE"]), int(c["ndofs"])
    vd = 3
    ess_s = np.asarray(c["ess_tdof"], np.int32)
    sc = DeviceOperator(c, ess=ess_s, layout=1 if capi.lib.mfem_amd_has_specialization(kind, D, Q) else 0)
    # vector operator: same essential dofs on every component
    ess_v = np.sort(np.concatenate([[vdof(i, k, nd, vd, byvdim) for i in ess_s] for k in range(vd)])).astype(np.int32)
    vec = DeviceOperator(c, layout=sc.layout)
    ess_d = dev(ess_v, np.int32)
    marker = zeros(nd*vd, torch.uint8)
    capi.call("mfem_amd_set_marker", len(ess_v), ptr(ess_d), ptr(marker), None)
    yE = zeros(NE*D**3*vd)
    vp = lambda a: a.ctypes.data_as(C.c_void_p)
    op = capi.PaOperator(kind, vec.layout, D, Q, NE, nd, vp(vec.B), vp(vec.G), ptr(vec.pa), ptr(vec.gm), None,
                         ptr(vec.off), ptr(vec.ind), len(ess_v), ptr(ess_d), ptr(marker), ptr(yE), vd, byvdim, 0)
    rng = np.random.default_rng(11)
    comps = [rng.random(nd) for _ in range(vd)]
    y, xin = zeros(nd*vd), dev(to_vec(comps, byvdim))
    capi.call("mfem_amd_pa_constrained_mult", C.byref(op), ptr(xin), ptr(y), None)
    for k, part in enumerate(split(host(y), nd, vd, byvdim)):
        xk = dev(comps[k])
        assert np.array_equal(part, host(sc.constrained_mult(xk))), (name, k)
    # PCG
    diag = np.asarray(c["diag"]).reshape(-1).copy()
    dinv_s = 1.0/diag
    dinv_s[ess_s] = 1.0
    b_s = np.asarray(c["B_rhs"]).reshape(-1)
    fac = [1.0, 0.5, 2.0]
    b_d, dinv_d = dev(b_s), dev(dinv_s)
    xs, rs, hs = sc.pcg(b_d, dinv_d, 1e-10, max_iter=500)
    n = nd*vd
    ws = zeros(capi.lib.mfem_amd_pcg_workspace_bytes(n, 500), torch.uint8)
    xv = zeros(n)
    res = capi.PcgResult()
    hist = np.zeros(502)
    dinv_v, b_v = dev(to_vec([dinv_s]*vd, byvdim)), dev(to_vec([b_s*f for f in fac], byvdim))
    capi.call("mfem_amd_pcg_pa", C.byref(op), ptr(dinv_v), ptr(b_v),
              ptr(xv), 1e-10, 0.0, 500, 1, ptr(ws), C.byref(res), hist.ctypes.data_as(C.c_void_p), None)
    assert (res.iterations, res.converged) == (rs.iterations, rs.converged), (res.iterations, rs.iterations)
    Xs = host(xs)
    for k, part in enumerate(split(host(xv), nd, vd, byvdim)):
        assert np.max(np.abs(part - fac[k]*Xs)) <= 1e-9*np.max(np.abs(Xs)), (name, k)
    # (B r, r) of the block system = (1 + 1/4 + 4) x the scalar one, up to the dot products' rounding
    assert abs(hist[0] - 5.25*hs[0]) <= 1e-12*hist[0]


@pytest.mark.parametrize("vd", [1, 2, 4, 5])
@pytest.mark.parametrize("byvdim", [0, 1])
def test_other_component_counts_and_edge_cases(vd, byvdim):
    """vdim = 1, 2, 4, 5 (the transpose-gather works three components at a time, the element kernel loops),
    the generic fallback kernel, an empty mesh, and the argument checks."""
    from gpu_helpers import DeviceOperator, dev, host, ptr, zeros
    from mfem_amd import capi
    capi.call("mfem_amd_init", 0)
    for name in ("cart2_bp3_p2", "fichera-q3_massvar_p2"):           # (3,4) specialised; (3,7) -> fallback kernel
        c = load_case(name)
        kind, D, Q, NE, nd = int(c["kind"]), int(c["D1D"]), int(c["Q1D"]), int(c["NE"]), int(c["ndofs"])
        op = DeviceOperator(c, layout=0)
        rng = np.random.default_rng(vd)
        comps = [rng.random(nd) for _ in range(vd)]
        x, yE, y = dev(to_vec(comps, byvdim)), zeros(NE*D**3*vd), zeros(nd*vd)
        capi.call("mfem_amd_pa_apply_L2E_vdim", kind, 0, D, Q, NE, vd, byvdim, nd, op.B, op.G, ptr(op.pa), 0,
                  ptr(op.gm), None, ptr(x), ptr(yE), None)
        capi.call("mfem_amd_restrict_scatter_vdim", nd, D**3, vd, byvdim, ptr(op.off), ptr(op.ind), ptr(yE), ptr(y), 0, 0, None)
        for k, part in enumerate(split(host(y), nd, vd, byvdim)):
            xk = dev(comps[k])
            assert np.array_equal(part, host(op.apply_L2L(xk))), (name, vd, k)
        # replicate: y[vdof(i,c)] = scale[c]*x[i]
        sc = np.arange(1.0, vd + 1.0)
        xs, yr = dev(comps[0]), zeros(nd*vd)
        capi.call("mfem_amd_vec_replicate_vdim", nd, vd, byvdim, sc.ctypes.data_as(C.c_void_p), ptr(xs), ptr(yr), None)
        for k, part in enumerate(split(host(yr), nd, vd, byvdim)):
            assert np.array_equal(part, sc[k]*comps[0])
    # empty mesh / empty space: no launch, no error
    capi.call("mfem_amd_pa_apply_L2E_vdim", 1, 0, 3, 4, 0, vd, byvdim, 0, op.B, op.G, ptr(op.pa), 0, ptr(op.gm), None, ptr(x), ptr(yE), None)
    capi.call("mfem_amd_restrict_scatter_vdim", 0, 27, vd, byvdim, ptr(op.off), ptr(op.ind), ptr(yE), ptr(y), 0, 0, None)
    with pytest.raises(capi.MfemAmdError, match="vdim"):
        capi.call("mfem_amd_pa_apply_L2E_vdim", kind, 0, D, Q, NE, 0, byvdim, nd, op.B, op.G, ptr(op.pa), 0, ptr(op.gm), None, ptr(x), ptr(yE), None)
    with pytest.raises(capi.MfemAmdError, match="add and abs"):
        capi.call("mfem_amd_restrict_scatter_vdim", nd, D**3, vd, byvdim, ptr(op.off), ptr(op.ind), ptr(yE), ptr(y), 1, 1, None)
