"""TEST INFRASTRUCTURE: an independent full assembly of the H1 mass / diffusion matrices on hexahedra,
written with dense numpy tensors (3D basis functions and physical gradients tabulated at every
quadrature point, element matrices, sparse sum over the gather map) -- no sum factorisation, no
partial-assembly data.  It plays the role of the reference's `AssemblyLevel::LEGACY` forms in the
equivalences its unit tests state (PA == assembled: tests/unit/fem/test_pa_kernels.cpp:641-694,
test_assembly_levels.cpp:96-196, test_pa_diagonal.cpp:92-141,182-278, test_pa_coeff.cpp,
test_operatorjacobismoother.cpp:17-84, tests/unit/mesh/test_geometric_factors.cpp:18-47)."""
import numpy as np
import scipy.sparse as sp


def tensor_basis(B, G, D, Q):
    """1D tables in MFEM layout [q + Q*d] -> phi (Q^3, D^3) and reference gradients (Q^3, D^3, 3);
    q = qx + Q*(qy + Q*qz), i = dx + D*(dy + D*dz)."""
    b, g = np.asarray(B).reshape(D, Q).T, np.asarray(G).reshape(D, Q).T          # [q, d]
    t = lambda X, Y, Z: np.einsum("ax,by,cz->cbazyx", X, Y, Z).reshape(Q**3, D**3)
    return t(b, b, b), np.stack([t(g, b, b), t(b, g, b), t(b, b, g)], axis=-1)


def geometry(nodes_E, Bn, Gn, Dn, Q, ne):
    """E-vector of the nodes (Dn^3, 3, NE) -> points (NE, Q^3, 3), J (NE, Q^3, 3, 3) with J[r, c] =
    d x_r / d xi_c, and det J."""
    N, dN = tensor_basis(Bn, Gn, Dn, Q)
    X = np.asarray(nodes_E).reshape(ne, 3, Dn**3)
    pts = np.einsum("ern,qn->eqr", X, N)
    J = np.einsum("ern,qnc->eqrc", X, dN)
    return pts, J, np.linalg.det(J)


def assemble(kind, D, Q, B, G, W, nodes_E, Bn, Gn, Dn, gather_map, ndofs, coeff=None, elements=None):
    """Sparse matrix of the mass (kind 0) or diffusion (kind 1) form.  coeff(pts (n,3)) -> (n,) scalar,
    (n,3) diagonal entries or (n,3,3) matrix K; a(u, v) = int K grad u . grad v, row = test function.
    elements: the subset the integrator acts on (attribute markers), default all."""
    ne = len(gather_map)//D**3
    gm = np.asarray(gather_map).reshape(ne, D**3)
    phi, dphi = tensor_basis(B, G, D, Q)
    pts, J, detJ = geometry(nodes_E, Bn, Gn, Dn, Q, ne)
    w = np.asarray(W).reshape(-1)
    rows, cols, vals = [], [], []
    for e in (range(ne) if elements is None else elements):
        c = np.ones(Q**3) if coeff is None else np.asarray(coeff(pts[e]))
        if kind == 0:
            Ae = np.einsum("q,qi,qj->ij", w*detJ[e]*c, phi, phi)
        else:
            K = c[:, None, None]*np.eye(3) if c.ndim == 1 else (np.einsum("qr,rs->qrs", c, np.eye(3)) if c.ndim == 2 else c)
            g = np.einsum("qic,qcr->qir", dphi, np.linalg.inv(J[e]))          # physical gradients
            Ae = np.einsum("q,qir,qrs,qjs->ij", w*detJ[e], g, K, g)
        rows.append(np.repeat(gm[e], D**3)); cols.append(np.tile(gm[e], D**3)); vals.append(Ae.reshape(-1))
    A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=(ndofs, ndofs))
    return A.tocsr()


# coefficient functions of the equivalence tests (own expressions; smooth, diffusion tensors positive definite)
def c_const(p):
    return np.full(len(p), 12.34)


def c_scalar(p):
    return 1.0 + p[:, 0]**2 + np.sin(2.0*p[:, 1])*p[:, 2]**2 + 0.5


def c_vector(p):
    return np.stack([1.0 + p[:, 0]**2, 2.0 + np.cos(p[:, 1]), 1.5 + p[:, 0]*p[:, 2]], axis=1)


def c_symmetric(p):
    K = np.zeros((len(p), 3, 3))
    K[:, 0, 0] = 2.0 + p[:, 0]**2; K[:, 1, 1] = 3.0 + np.sin(p[:, 1]); K[:, 2, 2] = 2.5 + p[:, 0]*p[:, 1]
    K[:, 0, 1] = K[:, 1, 0] = 0.3*p[:, 1]; K[:, 0, 2] = K[:, 2, 0] = 0.2*np.cos(3.0*p[:, 2]); K[:, 1, 2] = K[:, 2, 1] = -0.4*p[:, 2]
    return K


def c_matrix(p):
    K = c_symmetric(p)
    s1, s2, s3 = 0.7 + 0.1*p[:, 0], -0.5*p[:, 1], 0.25 + np.sin(p[:, 2])
    K[:, 0, 1] += s1; K[:, 1, 0] -= s1; K[:, 0, 2] += s2; K[:, 2, 0] -= s2; K[:, 1, 2] += s3; K[:, 2, 1] -= s3
    return K


COEFFS = {"const": (c_const, 1), "scalar": (c_scalar, 1), "vector": (c_vector, 3), "symmetric": (c_symmetric, 6), "matrix": (c_matrix, 9)}


def packed_coefficient(name, pts):
    """the coefficient at the points (NE, Q^3, 3) the way PADiffusionSetup3D takes it: (coeffDim, Q^3, NE)
    flattened, coeffDim 1 / 3 / 6 (upper triangle by rows) / 9 (row-major), or one number for a constant."""
    fn, cd = COEFFS[name]
    if name == "const":
        return np.array([12.34]), 1, 1
    ne, nq, _ = pts.shape
    v = np.asarray(fn(pts.reshape(-1, 3)))
    if cd == 6:
        v = np.stack([v[:, 0, 0], v[:, 0, 1], v[:, 0, 2], v[:, 1, 1], v[:, 1, 2], v[:, 2, 2]], axis=1)
    elif cd == 9:
        v = v.reshape(-1, 9)
    return np.ascontiguousarray(v.reshape(ne*nq*cd)), cd, 0
