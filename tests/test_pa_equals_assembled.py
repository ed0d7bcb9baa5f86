"""CPU: the equivalences the reference's unit tests state for this path, restated against an independent
dense full assembly (tests/full_assembly.py) with the oracle in the place of the PA form:
  * PA operator == assembled matrix, Mult and MultTranspose, every coefficient type
    (tests/unit/fem/test_pa_kernels.cpp:641-694, test_assembly_levels.cpp:96-196, test_pa_coeff.cpp)
  * PA diagonal == diagonal of the assembled matrix (test_pa_diagonal.cpp:92-141 mass, 182-278 diffusion with
    constant / function / vector / symmetric-matrix / matrix coefficients, orders 1..4, 1 and 8 elements)
  * OperatorJacobiSmoother on the PA diagonal == DSmoother of the eliminated matrix, DIAG_ONE
    (test_operatorjacobismoother.cpp:17-84)
  * geometric factors == pointwise element transformation (tests/unit/mesh/test_geometric_factors.cpp:18-47)
The same checks run on the device in tests/test_gpu_pa_equals_assembled.py."""
import numpy as np
import pytest

import full_assembly as fa
import oracle_lib as orc
from conftest import load_case

from mfem_amd.cartesian import CartesianProblem
from mfem_amd.tables import bakeoff_tables


def cartesian_setup(p, n, nonaligned=False):
    """ne = n^3 unit-cube mesh, H1 order p, the integrators' default rule (order 2p+2: p+2 Gauss-Legendre points).
    nonaligned: the vertices are moved by a skew + twist (the role of MakeCartesianNonaligned in
    tests/unit/fem/test_pa_coeff.cpp): trilinear elements that are neither axis-aligned nor affine."""
    prob = CartesianProblem(p, n, n, n)
    if nonaligned:
        X = np.asarray(prob.nodes_E).reshape(prob.ne, 3, 8).copy()
        x, y, z = X[:, 0].copy(), X[:, 1].copy(), X[:, 2].copy()
        X[:, 0] = x + 0.25*y + 0.1*z*z
        X[:, 1] = y + 0.15*np.sin(1.3*x) - 0.1*z
        X[:, 2] = z + 0.2*x*y
        prob.nodes_E = np.ascontiguousarray(X.reshape(-1))
    _, D, Q, B, G, Bn, Gn, W = bakeoff_tables("bp3", p)
    return prob, D, Q, B, G, Bn, Gn, W


def oracle_pa_data(prob, Q, Bn, Gn, W, C, cd, const_c):
    J = np.zeros(9*Q**3*prob.ne)
    orc.lib.orc_jacobians_3d(2, Q, prob.ne, Bn, Gn, prob.nodes_E, J)
    pa = np.zeros((9 if cd == 9 else 6)*Q**3*prob.ne)
    orc.lib.orc_diffusion_setup_3d(Q, cd, prob.ne, W, J, C, const_c, pa)
    return pa, J


@pytest.mark.parametrize("n,nonaligned", [(1, False), (2, False), (2, True)])
@pytest.mark.parametrize("p", [1, 2, 3, 4])
@pytest.mark.parametrize("ctype", list(fa.COEFFS))
def test_diffusion_pa_equals_assembled(p, n, nonaligned, ctype):
    prob, D, Q, B, G, Bn, Gn, W = cartesian_setup(p, n, nonaligned)
    A = fa.assemble(1, D, Q, B, G, W, prob.nodes_E, Bn, Gn, 2, prob.gather_map, prob.ndofs, fa.COEFFS[ctype][0])
    pts, _, _ = fa.geometry(prob.nodes_E, Bn, Gn, 2, Q, prob.ne)
    C, cd, const_c = fa.packed_coefficient(ctype, pts)
    pa, _ = oracle_pa_data(prob, Q, Bn, Gn, W, C, cd, const_c)
    op = orc.Operator(2 if cd == 9 else 1, D, Q, prob.ne, prob.ndofs, B, G, pa, prob.gather_map, prob.offsets, prob.indices)
    x = np.random.default_rng(1).random(prob.ndofs)
    scale = abs(A).max()
    assert np.linalg.norm(op.mult(x) - A @ x) < 1e-12*scale*np.sqrt(prob.ndofs)
    assert np.linalg.norm(op.diag() - A.diagonal()) < 1e-12*scale*np.sqrt(prob.ndofs)
    if cd != 9:        # symmetric D: MultTranspose is Mult (bilininteg_diffusion_pa.cpp:76-87)
        assert np.linalg.norm(op.mult(x) - A.T @ x) < 1e-12*scale*np.sqrt(prob.ndofs)
    else:
        assert np.linalg.norm((A - A.T) @ x) > 1e-3*np.linalg.norm(A @ x)        # really non-symmetric


@pytest.mark.parametrize("p", [1, 2, 3, 4])
def test_mass_pa_equals_assembled(p):
    prob = CartesianProblem(p, 3, 3, 3)
    _, D, Q, B, G, Bn, Gn, W = bakeoff_tables("bp1", p)
    A = fa.assemble(0, D, Q, B, G, W, prob.nodes_E, Bn, Gn, 2, prob.gather_map, prob.ndofs, fa.c_scalar)
    pts, _, detJ = fa.geometry(prob.nodes_E, Bn, Gn, 2, Q, prob.ne)
    pa = np.zeros(Q**3*prob.ne)
    orc.lib.orc_mass_setup_3d(Q, prob.ne, W, np.ascontiguousarray(detJ.reshape(-1)), np.ascontiguousarray(fa.c_scalar(pts.reshape(-1, 3))), 0, pa)
    op = orc.Operator(0, D, Q, prob.ne, prob.ndofs, B, G, pa, prob.gather_map, prob.offsets, prob.indices)
    x = np.random.default_rng(2).random(prob.ndofs)
    assert np.linalg.norm(op.mult(x) - A @ x) < 1e-12*np.linalg.norm(A @ x)
    assert np.linalg.norm(op.diag() - A.diagonal()) < 1e-12*np.linalg.norm(A.diagonal())


@pytest.mark.parametrize("n", [1, 2])
@pytest.mark.parametrize("p", [1, 2, 3, 4])
def test_jacobi_smoother_equals_dsmoother_of_eliminated_matrix(p, n):
    prob, D, Q, B, G, Bn, Gn, W = cartesian_setup(p, n)
    A = fa.assemble(1, D, Q, B, G, W, prob.nodes_E, Bn, Gn, 2, prob.gather_map, prob.ndofs).tolil()
    ess = prob.boundary_dofs()
    d_elim = A.diagonal().copy()
    d_elim[ess] = 1.0                                   # FormLinearSystem with Matrix::DIAG_ONE
    pa, _ = oracle_pa_data(prob, Q, Bn, Gn, W, np.ones(1), 1, 1)
    op = orc.Operator(1, D, Q, prob.ne, prob.ndofs, B, G, pa, prob.gather_map, prob.offsets, prob.indices)
    dinv = orc.jacobi_setup(op.diag(), ess)
    x = np.random.default_rng(3).random(prob.ndofs)
    assert np.linalg.norm(dinv*x - x/d_elim) < 1e-12*np.linalg.norm(x/d_elim)


@pytest.mark.parametrize("name", ["fichera-q3_diff_p2", "toroid-hex_diff_p2"])
def test_curved_mesh_pa_and_geometric_factors(name):
    """the reference's curved-mesh cases (fichera-q3 in test_pa_kernels.cpp / test_geometric_factors.cpp): tables of
    the fixture, dense assembly on its order-3 nodes"""
    c = load_case(name)
    D, Q, NE, Dn = int(c["D1D"]), int(c["Q1D"]), int(c["NE"]), int(c["nodes_D1D"])
    pts, J, detJ = fa.geometry(c["nodes_E"], c["nodes_B"], c["nodes_G"], Dn, Q, NE)
    Jo = np.zeros(9*Q**3*NE)
    orc.lib.orc_jacobians_3d(Dn, Q, NE, c["nodes_B"], c["nodes_G"], c["nodes_E"], Jo)
    Jo = Jo.reshape(NE, 3, 3, Q**3).transpose(0, 3, 2, 1)          # J(q, r, c, e) -> [e, q, r, c]
    assert np.max(np.abs(Jo - J)) < 1e-12*np.max(np.abs(J))
    do = np.zeros(Q**3*NE)
    orc.lib.orc_det_3d(Q, NE, np.ascontiguousarray(Jo.transpose(0, 3, 2, 1)).reshape(-1), do)
    assert np.max(np.abs(do.reshape(NE, -1) - detJ)) < 1e-12*np.max(np.abs(detJ))
    assert np.max(np.abs(np.asarray(c["detJ"]).reshape(NE, -1) - detJ)) < 1e-12*np.max(np.abs(detJ))      # the reference's own detJ
    A = fa.assemble(1, D, Q, c["B"], c["G"], c["W"], c["nodes_E"], c["nodes_B"], c["nodes_G"], Dn, c["gather_map"], int(c["ndofs"]))
    assert np.linalg.norm(A @ c["x"] - c["y"]) < 1e-11*np.linalg.norm(c["y"])                  # the reference's PA result
    assert np.linalg.norm(A.diagonal() - c["diag"]) < 1e-11*np.linalg.norm(c["diag"])
