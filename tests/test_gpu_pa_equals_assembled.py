"""-m gpu: the reference's unit-test equivalences (PA == assembled; see tests/test_pa_equals_assembled.py for the
list with file:line) with the device path through the C ABI in the place of the PA form, against the
independent dense assembly of tests/full_assembly.py.  Tolerance 1e-12 relative, as the reference's tests."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

import full_assembly as fa                                                        # noqa: E402
from conftest import load_case, splitmix                                                    # noqa: E402
from test_pa_equals_assembled import cartesian_setup                              # noqa: E402


def device_operator(kind, D, Q, ne, ndofs, B, G, Bn, Gn, Dn, W, nodes_E, gather_map, offsets, indices, C, cd, const_c, x):
    """AssemblePA + Mult + AssembleDiagonal through the C ABI; returns y = A x and diag(A) on the host"""
    from gpu_helpers import dev, hB, host, ptr, zeros
    from mfem_amd import capi
    capi.call("mfem_amd_init", 0)
    nonsym = cd == 9
    okind = 2 if nonsym else kind
    layouts = [0] + ([1] if (not nonsym and capi.lib.mfem_amd_has_specialization(kind, D, Q)) else [])
    Wd, nE, Cd = dev(W), dev(nodes_E), dev(C)
    gm, off, ind, xd = dev(gather_map, np.int32), dev(offsets, np.int32), dev(indices, np.int32), dev(x)
    out = []
    for layout in layouts:
        pa = zeros(capi.lib.mfem_amd_pa_data_size(okind, layout, Q, ne))
        capi.call("mfem_amd_pa_setup_from_nodes_3d", kind, Dn, Q, cd, ne, hB(Bn), hB(Gn), ptr(Wd), ptr(nE), ptr(Cd), const_c,
                  layout, ptr(pa), None, None)
        yE, y, dE, dg = zeros(ne*D**3), zeros(ndofs), zeros(ne*D**3), zeros(ndofs)
        capi.call("mfem_amd_pa_apply_L2E", okind, layout, D, Q, ne, hB(B), hB(G), ptr(pa), ptr(gm), ptr(xd), ptr(yE), None)
        capi.call("mfem_amd_restrict_scatter", ndofs, ptr(off), ptr(ind), ptr(yE), ptr(y), 0, 0, None)
        if kind:
            capi.call("mfem_amd_pa_diffusion_diag_3d", D, Q, ne, 0 if nonsym else 1, hB(B), hB(G), ptr(pa), layout, ptr(dE), None)
        else:
            capi.call("mfem_amd_pa_mass_diag_3d", D, Q, ne, hB(B), ptr(pa), layout, ptr(dE), None)
        capi.call("mfem_amd_restrict_scatter", ndofs, ptr(off), ptr(ind), ptr(dE), ptr(dg), 0, 1, None)
        out.append((host(y), host(dg)))
    return out


@pytest.mark.parametrize("n,nonaligned", [(1, False), (2, False), (2, True)])
@pytest.mark.parametrize("p", [1, 2, 3, 4])
@pytest.mark.parametrize("ctype", list(fa.COEFFS))
def test_diffusion_pa_equals_assembled(p, n, nonaligned, ctype):
    prob, D, Q, B, G, Bn, Gn, W = cartesian_setup(p, n, nonaligned)
    A = fa.assemble(1, D, Q, B, G, W, prob.nodes_E, Bn, Gn, 2, prob.gather_map, prob.ndofs, fa.COEFFS[ctype][0])
    pts, _, _ = fa.geometry(prob.nodes_E, Bn, Gn, 2, Q, prob.ne)
    C, cd, const_c = fa.packed_coefficient(ctype, pts)
    x = np.random.default_rng(1).random(prob.ndofs)
    tol = 1e-12*abs(A).max()*np.sqrt(prob.ndofs)
    for y, dg in device_operator(1, D, Q, prob.ne, prob.ndofs, B, G, Bn, Gn, 2, W, prob.nodes_E, prob.gather_map, prob.offsets,
                                 prob.indices, C, cd, const_c, x):
        assert np.linalg.norm(y - A @ x) < tol
        assert np.linalg.norm(dg - A.diagonal()) < tol
        if cd != 9:
            assert np.linalg.norm(y - A.T @ x) < tol


@pytest.mark.parametrize("p", [1, 2, 3, 4])
def test_mass_pa_equals_assembled(p):
    from mfem_amd.cartesian import CartesianProblem
    from mfem_amd.tables import bakeoff_tables
    prob = CartesianProblem(p, 3, 3, 3)
    _, D, Q, B, G, Bn, Gn, W = bakeoff_tables("bp1", p)
    A = fa.assemble(0, D, Q, B, G, W, prob.nodes_E, Bn, Gn, 2, prob.gather_map, prob.ndofs, fa.c_scalar)
    pts, _, _ = fa.geometry(prob.nodes_E, Bn, Gn, 2, Q, prob.ne)
    C = np.ascontiguousarray(fa.c_scalar(pts.reshape(-1, 3)))
    x = np.random.default_rng(2).random(prob.ndofs)
    for y, dg in device_operator(0, D, Q, prob.ne, prob.ndofs, B, G, Bn, Gn, 2, W, prob.nodes_E, prob.gather_map, prob.offsets,
                                 prob.indices, C, 1, 0, x):
        assert np.linalg.norm(y - A @ x) < 1e-12*np.linalg.norm(A @ x)
        assert np.linalg.norm(dg - A.diagonal()) < 1e-12*np.linalg.norm(A.diagonal())


@pytest.mark.parametrize("n", [1, 2])
@pytest.mark.parametrize("p", [1, 2, 3, 4])
def test_jacobi_smoother_equals_dsmoother_of_eliminated_matrix(p, n):
    from gpu_helpers import dev, host, ptr, zeros
    from mfem_amd import capi
    prob, D, Q, B, G, Bn, Gn, W = cartesian_setup(p, n)
    A = fa.assemble(1, D, Q, B, G, W, prob.nodes_E, Bn, Gn, 2, prob.gather_map, prob.ndofs)
    ess = prob.boundary_dofs().astype(np.int32)
    d_elim = A.diagonal().copy()
    d_elim[ess] = 1.0
    x = np.random.default_rng(3).random(prob.ndofs)
    (_, dg), = device_operator(1, D, Q, prob.ne, prob.ndofs, B, G, Bn, Gn, 2, W, prob.nodes_E, prob.gather_map, prob.offsets,
                               prob.indices, np.ones(1), 1, 1, x)[:1]
    dgd, essd, xd = dev(dg), dev(ess, np.int32), dev(x)
    dinv, z = zeros(prob.ndofs), zeros(prob.ndofs)
    capi.call("mfem_amd_jacobi_setup", prob.ndofs, ptr(dgd), 1.0, len(ess), ptr(essd), ptr(dinv), None)
    capi.call("mfem_amd_jacobi_apply", prob.ndofs, ptr(dinv), ptr(xd), ptr(z), None)
    assert np.linalg.norm(host(z) - x/d_elim) < 1e-12*np.linalg.norm(x/d_elim)


@pytest.mark.parametrize("name", ["fichera-q3_diff_p2", "toroid-hex_diff_p2"])
def test_curved_mesh_pa_and_geometric_factors(name):
    from gpu_helpers import dev, hB, host, ptr, zeros
    from mfem_amd import capi
    capi.call("mfem_amd_init", 0)
    c = load_case(name)
    D, Q, NE, Dn, nd = int(c["D1D"]), int(c["Q1D"]), int(c["NE"]), int(c["nodes_D1D"]), int(c["ndofs"])
    _, J, detJ = fa.geometry(c["nodes_E"], c["nodes_B"], c["nodes_G"], Dn, Q, NE)
    nE, Jd, dd = dev(c["nodes_E"]), zeros(9*Q**3*NE), zeros(Q**3*NE)
    capi.call("mfem_amd_geom_jacobians_3d", Dn, Q, NE, hB(c["nodes_B"]), hB(c["nodes_G"]), ptr(nE), ptr(Jd), None)
    capi.call("mfem_amd_geom_det_3d", Q, NE, ptr(Jd), ptr(dd), None)
    Jh = host(Jd).reshape(NE, 3, 3, Q**3).transpose(0, 3, 2, 1)
    assert np.max(np.abs(Jh - J)) < 1e-12*np.max(np.abs(J))
    assert np.max(np.abs(host(dd).reshape(NE, -1) - detJ)) < 1e-12*np.max(np.abs(detJ))
    A = fa.assemble(1, D, Q, c["B"], c["G"], c["W"], c["nodes_E"], c["nodes_B"], c["nodes_G"], Dn, c["gather_map"], nd)
    for y, dg in device_operator(1, D, Q, NE, nd, c["B"], c["G"], c["nodes_B"], c["nodes_G"], Dn, c["W"], c["nodes_E"],
                                 c["gather_map"], c["offsets"], c["indices"], np.ones(1), 1, 1, c["x"]):
        assert np.linalg.norm(y - A @ c["x"]) < 1e-11*np.linalg.norm(A @ c["x"])
        assert np.linalg.norm(dg - A.diagonal()) < 1e-11*np.linalg.norm(A.diagonal())


# ---- element-attribute markers (tests/unit/fem/test_pa_kernels.cpp:696-746 "PA Markers") ---------------
def test_marker_kernels_select_elements():
    from gpu_helpers import dev, host, ptr
    from mfem_amd import capi
    capi.call("mfem_amd_init", 0)
    rng = np.random.default_rng(4)
    ne, nd = 37, 27
    attr = rng.integers(1, 4, ne).astype(np.int32)
    attr[5] = 0                                              # non-positive attribute: never marked (:768)
    markers = np.array([1, 0, 1], np.int32)
    x, y = rng.random(ne*nd), rng.random(ne*nd)
    on = np.array([a > 0 and markers[a - 1] != 0 for a in attr])
    xd, yd, dd = dev(x), dev(y), dev(y)
    md, ad = dev(markers, np.int32), dev(attr, np.int32)
    capi.call("mfem_amd_evec_add_marked", ne, nd, 3, ptr(md), ptr(ad), ptr(xd), ptr(yd), None)
    want = y.reshape(ne, nd) + np.where(on[:, None], x.reshape(ne, nd), 0.0)
    assert np.array_equal(host(yd).reshape(ne, nd), want)
    capi.call("mfem_amd_evec_zero_unmarked", ne, nd, 3, ptr(md), ptr(ad), ptr(dd), None)
    assert np.array_equal(host(dd).reshape(ne, nd), np.where(on[:, None], y.reshape(ne, nd), 0.0))


@pytest.mark.parametrize("kind,p", [(0, 2), (1, 2), (1, 3)])
def test_cxx_forms_with_markers_equal_assembled(kind, p, tmp_path):
    """BilinearForm::AddDomainIntegrator(bfi, marker) of the C++ layer: Mult, MultTranspose (symmetric forms) and
    AssembleDiagonal equal the dense assembly over the marked elements (attributes 1 + e%2, marker {0, 1})"""
    import subprocess
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    spec = ["cart", "3", "2", "2", "1", "1", "1", "1", str(p), "0", str(kind), "0", "run", "markers"]
    out = subprocess.run([str(root / "examples/bin/dump_tables"), str(tmp_path)] + spec, capture_output=True, text=True, timeout=300, cwd=str(root))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    r = {f.stem: np.fromfile(f, dtype=np.int32 if f.suffix == ".i32" else np.float64) for f in tmp_path.iterdir()}
    # the tables the C++ layer built (MFEM's dof numbering and space-filling element order)
    gm, B, G, W, Bn, Gn = r["gather_map"], r["B"], r["G"], r["W"], r["nodes_B"], r["nodes_G"]
    Q = round(len(W)**(1/3)); D = len(B)//Q; Dn = len(Bn)//Q
    ne, ndofs = len(gm)//D**3, len(r["y"])
    ngm, nn = r["nodes_gather_map"].reshape(ne, Dn**3), len(r["nodes"])//3
    comp = (lambda g, c: r["nodes"][3*g + c]) if r["nodes_byvdim"][0] else (lambda g, c: r["nodes"][g + c*nn])
    nodes_E = np.stack([comp(ngm, c) for c in range(3)], axis=1).reshape(-1)                  # (Dn^3, 3, NE)
    A = fa.assemble(kind, D, Q, B, G, W, nodes_E, Bn, Gn, Dn, gm, ndofs, elements=range(1, ne, 2))
    x = splitmix(1, ndofs)
    assert np.linalg.norm(r["y"] - A @ x) < 1e-12*np.linalg.norm(A @ x)
    assert np.linalg.norm(r["diag"] - A.diagonal()) < 1e-12*np.linalg.norm(A.diagonal())
    untouched = np.setdiff1d(np.arange(ndofs), gm.reshape(ne, -1)[1::2].reshape(-1))
    assert len(untouched) and np.all(r["y"][untouched] == 0.0)

