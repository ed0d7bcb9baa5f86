"""Synthetic CEED bake-off inputs (host side, integer/numpy plumbing only).

Builds, for a box of nx*ny*nz hexahedra and H1 order p, the index arrays the PA path
consumes: the lexicographic element->L-dof ``gather_map``, its transpose ``offsets`` /
``indices`` in the reference's order (element-major, then local index:
fem/restriction.cpp:86-99), the essential (boundary) dof list and the trilinear E-nodes.
L-dofs are numbered lexicographically on the (p*nx+1)(p*ny+1)(p*nz+1) tensor grid and
elements lexicographically (x fastest) -- the reference's Make3D default is a
space-filling-curve element order with vertex/edge/face/interior dof numbering
(mesh/mesh.cpp:3750-3775, fem/fespace.cpp:3392-3500); the operator is the same up to a
permutation, see mfem_amd/host for the generator that reproduces MFEM's numbering.
A rank of a box-partitioned run passes its sub-box through ``origin``/``global_n``.
"""
import numpy as np


class CartesianProblem:
    def __init__(self, p, nx, ny, nz, sx=1.0, sy=1.0, sz=1.0, origin=(0, 0, 0), global_n=None, transpose=True):
        self.p, self.n = p, (nx, ny, nz)
        gn = global_n or (nx, ny, nz)
        D = p + 1
        n1 = (p*nx + 1, p*ny + 1, p*nz + 1)
        self.n1 = n1
        self.ndofs = n1[0]*n1[1]*n1[2]
        self.ne = nx*ny*nz
        assert self.ne*D**3 < 2**31 and self.ndofs < 2**31, "int32 index arrays (as the reference's)"
        self.offsets = self.indices = None
        ex, ey, ez = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
        # element index e = ex + nx*(ey + ny*ez)  -> order arrays with ez slowest
        ex, ey, ez = [a.transpose(2, 1, 0).reshape(-1) for a in (ex, ey, ez)]
        d = np.arange(D)
        gx = (ex[:, None]*p + d[None, :])                      # (ne, D)
        gy = (ey[:, None]*p + d[None, :])
        gz = (ez[:, None]*p + d[None, :])
        g = (gx[:, None, None, :] + n1[0]*(gy[:, None, :, None] + n1[1]*gz[:, :, None, None]))
        self.gather_map = np.ascontiguousarray(g.reshape(-1), np.int32)   # (e, dz, dy, dx)
        if transpose:
            # transpose table: stable sort by L-dof keeps (element, local) order within a dof
            order = np.argsort(self.gather_map, kind="stable")
            self.indices = order.astype(np.int32)
            counts = np.bincount(self.gather_map, minlength=self.ndofs)
            self.offsets = np.zeros(self.ndofs + 1, np.int32)
            np.cumsum(counts, out=self.offsets[1:])
        # trilinear E-nodes (Dn = 2): (2,2,2,3,ne), x fastest
        hx, hy, hz = sx/gn[0], sy/gn[1], sz/gn[2]
        X = np.empty((self.ne, 3, 2, 2, 2))
        for c, (ecoord, h, o) in enumerate(((ex, hx, origin[0]), (ey, hy, origin[1]), (ez, hz, origin[2]))):
            lo = (ecoord + o)*h
            corner = np.zeros((2, 2, 2))
            idx = np.arange(2)
            shape = [1, 1, 1]; shape[2 - c] = 2
            corner = corner + idx.reshape(shape)
            X[:, c] = lo[:, None, None, None] + h*corner[None]
        self.nodes_E = np.ascontiguousarray(X.reshape(-1))
        self.origin, self.global_n = origin, gn

    def boundary_dofs(self):
        """Essential dofs: L-dofs on the faces of the GLOBAL box (all 6 faces Dirichlet,
        tests/benchmarks/bench_ceed.cpp:99-100)."""
        p, n1 = self.p, self.n1
        o, gn, n = self.origin, self.global_n, self.n
        ix, iy, iz = np.arange(n1[0]), np.arange(n1[1]), np.arange(n1[2])
        bx = ((ix == 0) & (o[0] == 0)) | ((ix == n1[0] - 1) & (o[0] + n[0] == gn[0]))
        by = ((iy == 0) & (o[1] == 0)) | ((iy == n1[1] - 1) & (o[1] + n[1] == gn[1]))
        bz = ((iz == 0) & (o[2] == 0)) | ((iz == n1[2] - 1) & (o[2] + n[2] == gn[2]))
        m = bz[:, None, None] | by[None, :, None] | bx[None, None, :]
        return np.flatnonzero(m.reshape(-1)).astype(np.int32)
