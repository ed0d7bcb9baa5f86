"""Host-side 1D tables for the drivers (numpy): Gauss-Legendre / Gauss-Lobatto rules on [0,1]
and Lagrange basis tables in MFEM's DofToQuad::TENSOR layout B[q + Q*d], G[q + Q*d]
(fem/fe/fe_base.cpp:2622-2673; rules fem/intrules.cpp:424-642).  Checked against the
reference's own tables in tests/test_host_tables.py (agreement ~1e-15)."""
import numpy as np


def gauss_legendre(n):
    x, w = np.polynomial.legendre.leggauss(n)
    return 0.5*(x + 1.0), 0.5*w


def gauss_lobatto(n):
    if n == 1:
        return np.array([0.5]), np.array([1.0])
    c = np.zeros(n); c[-1] = 1.0
    P = np.polynomial.legendre.Legendre(c)
    r = P.deriv().roots() if n > 2 else np.zeros(0)
    x = np.concatenate(([-1.0], np.sort(r.real), [1.0]))
    w = 2.0/(n*(n - 1)*P(x)**2)
    return 0.5*(x + 1.0), 0.5*w


def lagrange_tables(nodes, pts):
    n, Q = len(nodes), len(pts)
    B, G = np.zeros((n, Q)), np.zeros((n, Q))
    for d in range(n):
        others = np.delete(nodes, d)
        denom = np.prod(nodes[d] - others)
        for q, y in enumerate(pts):
            B[d, q] = np.prod(y - others)/denom
            G[d, q] = sum(np.prod(np.delete(y - others, k)) for k in range(n - 1))/denom
    return np.ascontiguousarray(B.reshape(-1)), np.ascontiguousarray(G.reshape(-1))


def bakeoff_tables(problem, p):
    """(kind, D1D, Q1D, B, G, Bn, Gn, W) of CEED BP1/BP3 (Gauss-Legendre, Q = p+2) and BP5
    (Gauss-Lobatto, Q = p+1), tests/benchmarks/bench_ceed.cpp:45-51,144-160."""
    # bp2 / bp4 / bp6 are the vector (vdim = 3) versions of bp1 / bp3 / bp5: same tables
    problem = {"bp2": "bp1", "bp4": "bp3", "bp6": "bp5"}.get(problem, problem)
    kind = 0 if problem == "bp1" else 1
    D = p + 1
    if problem == "bp5":
        Q = p + 1
        qx, qw = gauss_lobatto(Q)
    else:
        Q = p + 2
        qx, qw = gauss_legendre(Q)
    nodes, _ = gauss_lobatto(D)
    B, G = lagrange_tables(nodes, qx)
    Bn, Gn = lagrange_tables(np.array([0.0, 1.0]), qx)
    W = np.ascontiguousarray((qw[None, None, :]*qw[None, :, None]*qw[:, None, None]).reshape(-1))
    return kind, D, Q, B, G, Bn, Gn, W
