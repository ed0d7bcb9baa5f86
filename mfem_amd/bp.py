"""CEED bake-off problem driver on the C ABI (device-resident), the Python twin of
tests/benchmarks/bench_ceed.cpp `BakeOff`/`Problem`/`Kernel`: Cartesian hex mesh, H1 order
p, BP1 (mass) / BP3 (diffusion, GL) / BP5 (diffusion, GLL) and their vector versions BP2 / BP4 /
BP6 (vdim = 3, Ordering::byVDIM as bench_ceed.cpp:47; in parallel the Exchange is built with vdim = 3), all-boundary Dirichlet,
Jacobi-PCG.  Serial, or one box per rank with the exchange of mfem_amd/parallel.py.
torch supplies device memory and the stream; every computation is a HIP kernel of
libmfem_amd.so."""
import ctypes as C
import os

import numpy as np
import torch

from . import capi
from .cartesian import CartesianProblem
from .tables import bakeoff_tables


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class BakeoffProblem:
    def __init__(self, problem, p, n, device, part=None, exchange=None, layout=1, device_tables=False,
                 numbering="lexicographic"):
        """`part`: None (serial Cartesian box of n elements), a BoxPartition, or a GeneralPartition of any hex
        mesh's tables (then n is ignored).  device_tables: build the E->L table (`offsets`, `indices`) from the
        gather map on the device (a stable sort by L-dof = the reference's (element, local index) order within a
        dof, fem/restriction.cpp:86-99) instead of numpy on the host -- the same integers, seconds instead of a
        minute at 1e8 dofs."""
        self.dev = torch.device(device)
        self.part, self.ex, self.layout = part, exchange, layout
        if part is not None:
            self.prob = part.prob
        elif numbering == "mfem":
            # the reference's element order and dof numbering (MakeCartesian3D + H1 space, built by the C++ host layer)
            from .hosttables import MfemBox
            self.prob = MfemBox(p, n[0], n[1], n[2], transpose=not device_tables)
        else:
            self.prob = CartesianProblem(p, n[0], n[1], n[2], transpose=not device_tables)
        pr = self.prob
        self.kind, self.D, self.Q, self.B, self.G, Bn, Gn, W = bakeoff_tables(problem, p)
        kind, D, Q = self.kind, self.D, self.Q
        self.vdim = 3 if problem in ("bp2", "bp4", "bp6") else 0
        vd = max(1, self.vdim)
        # nodal space of the mesh: trilinear for the synthetic boxes, order Dn-1 for mesh files
        self.Dn = getattr(pr, "Dn", 2)
        if self.Dn != 2:
            from .tables import gauss_legendre, gauss_lobatto, lagrange_tables
            qx, _ = gauss_lobatto(Q) if problem in ("bp5", "bp6") else gauss_legendre(Q)
            Bn, Gn = lagrange_tables(gauss_lobatto(self.Dn)[0], qx)
        self.ne, self.nscalar = pr.ne, pr.ndofs
        self.ndofs = pr.ndofs*vd                      # unknowns (vdofs of a vector problem)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        self.stream = C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)
        gather_map, offsets, indices, nodes_E = pr.gather_map, pr.offsets, pr.indices, pr.nodes_E
        # Exchange/compute overlap: elements that touch no ghost dof come first, so that their
        # part of the apply runs while the owner -> ghost exchange is in flight on a second stream.
        self.e_split = 0
        if part is not None and exchange is not None and os.environ.get("MFEM_AMD_OVERLAP", "1") != "0":
            gmap = np.asarray(gather_map).reshape(self.ne, D**3)
            iface = np.asarray(part.ghost, bool)[gmap].any(axis=1)
            n_int = int((~iface).sum())
            # the blocked pa_data layout is addressed per batch of NBZ elements: split on a batch boundary
            nbz = capi.lib.mfem_amd_pa_data_size(kind, 1, Q, 1)//((6 if kind else 1)*Q**3) if layout else 1
            self.e_split = (n_int//nbz)*nbz
            ordered = not iface[:n_int].any()          # the partition already lists its interior elements first
            if 0 < self.e_split and iface.any() and ordered:
                pass
            elif 0 < self.e_split and iface.any():
                perm = np.concatenate([np.flatnonzero(~iface), np.flatnonzero(iface)])
                gmap = gmap[perm]
                nodes_E = np.asarray(nodes_E).reshape(self.ne, -1)[perm].reshape(-1)
                gather_map = np.ascontiguousarray(gmap.reshape(-1))
                # E -> L table of the permuted order: per dof, ascending E-vector position (restriction.cpp:86-99)
                indices = np.argsort(gather_map, kind="stable").astype(np.int32)
                offsets = np.zeros(pr.ndofs + 1, np.int32)
                np.cumsum(np.bincount(gather_map, minlength=pr.ndofs), out=offsets[1:])
            else:
                self.e_split = 0
        self._nodes_E = nodes_E
        self.gm = t(gather_map)
        # numbering = "interior-last": vertex/edge/face dofs first (in their lexicographic order), then the element-
        # interior dofs element by element -- the class of numberings the reference gives every H1 space
        # (fem/fespace.cpp) -- so that mfem_amd_interior_rows applies.  self.perm[old] = new.
        self.perm = None
        if numbering == "interior-last" and part is None and D >= 3 and not self.vdim:
            ni = (D - 2)**3
            inter = self.gm.view(self.ne, D, D, D)[:, 1:-1, 1:-1, 1:-1].reshape(-1).long()
            is_int = torch.zeros(pr.ndofs, dtype=torch.bool, device=self.dev)
            is_int[inter] = True
            i0 = pr.ndofs - self.ne*ni
            perm = torch.empty(pr.ndofs, dtype=torch.int64, device=self.dev)
            perm[~is_int] = torch.arange(i0, device=self.dev)
            perm[inter] = i0 + torch.arange(self.ne*ni, device=self.dev)
            self.gm = perm[self.gm.long()].to(torch.int32)
            self.perm = perm
            offsets = None
            del inter, is_int
        if offsets is None:
            ind = torch.sort(self.gm, stable=True).indices
            self.ind = ind.to(torch.int32)
            del ind
            self.off = torch.zeros(pr.ndofs + 1, dtype=torch.int32, device=self.dev)
            self.off[1:] = torch.cumsum(torch.bincount(self.gm, minlength=pr.ndofs), 0).to(torch.int32)
        else:
            self.off, self.ind = t(offsets), t(indices)
        nodesE, Wd, one = t(nodes_E), t(W), torch.ones(1, dtype=torch.float64, device=self.dev)
        self.pa = torch.zeros(capi.lib.mfem_amd_pa_data_size(kind, layout, Q, self.ne), dtype=torch.float64, device=self.dev)
        capi.call("mfem_amd_pa_setup_from_nodes_3d", kind, self.Dn, Q, 1, self.ne, Bn, Gn, _p(Wd), _p(nodesE), _p(one), 1,
                  layout, _p(self.pa), None, self.stream)
        self.yE = torch.empty(self.ne*D**3*vd, dtype=torch.float64, device=self.dev)
        # essential dofs: all six faces of the global box (bench_ceed.cpp:99-100), every component
        if part is not None:
            marker, ess = part.essential_marker()
        else:
            ess = pr.boundary_dofs()
            if self.perm is not None:
                ess = self.perm[torch.from_numpy(np.asarray(ess, np.int64)).to(self.dev)].cpu().numpy()
            marker = np.zeros(self.nscalar, np.uint8); marker[ess] = 1
        if self.vdim:
            ess = (ess.astype(np.int64)[:, None]*vd + np.arange(vd)[None, :]).reshape(-1)      # byVDIM vdofs
            marker = np.repeat(marker, vd)
        self.ess_host = ess
        self.ess, self.marker = t(ess.astype(np.int32)), t(marker)
        self.masked = None
        if not self.vdim:       # vector spaces apply the marker in the gather instead
            self.masked = torch.empty(self.ne*D**3, dtype=torch.int32, device=self.dev)
            capi.call("mfem_amd_mask_gather_map", self.ne*D**3, _p(self.gm), _p(self.marker), _p(self.masked), self.stream)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        self.op = capi.PaOperator(kind, layout, D, Q, self.ne, self.nscalar, vp(self.B), vp(self.G), _p(self.pa),
                                  _p(self.gm), _p(self.masked), _p(self.off), _p(self.ind), len(ess), _p(self.ess),
                                  _p(self.marker), _p(self.yE), self.vdim, 1 if self.vdim else 0, 0)
        # element-interior rows the element kernel may write itself (include/mfem_amd.h: mfem_amd_interior_rows)
        self.direct = 0
        if not self.vdim and os.environ.get("MFEM_AMD_DIRECT_ROWS", "1") != "0":
            work = torch.zeros(1, dtype=torch.int32, device=self.dev)
            dr = C.c_int64(0)
            capi.call("mfem_amd_interior_rows", D, self.ne, self.nscalar, _p(self.gm), _p(self.off), _p(work), C.byref(dr), self.stream)
            self.direct = int(dr.value)
        self.op.direct_rows = self.direct
        # with them, the E-vector in its packed form: `indices` remapped once (include/mfem_amd.h: mfem_amd_pack_indices)
        self.pind = None
        if self.direct:
            self.pind = torch.empty_like(self.ind)
            capi.call("mfem_amd_pack_indices", D, self.ind.numel(), _p(self.ind), _p(self.pind), self.stream)
            self.op.packed_indices = self.pind.data_ptr()
        # the library overlaps the owner -> ghost exchange with the first e_split elements (include/mfem_amd.h)
        self.op.e_split = 0 if self.vdim else self.e_split
        self.ghost = None
        if part is not None:
            g = np.asarray(part.ext_ldof, np.int64)
            if self.vdim:
                g = (g[:, None]*vd + np.arange(vd)[None, :]).reshape(-1)
            self.ghost = t(g.astype(np.int32))
            # ... and the ghost -> owner exchange with the transpose-gather: the ghost rows are summed first (include/mfem_amd.h)
            if not self.vdim:
                self.op.reverse_overlap, self.op.n_ghost_rows = 1, self.ghost.numel()
                self.op.ghost_rows = self.ghost.data_ptr() if self.ghost.numel() else None
        # BilinearForm::Mult is the unconstrained operator: the same tables without the essential list
        self.op_free = capi.PaOperator.from_buffer_copy(self.op)
        self.op_free.ness, self.op_free.ess, self.op_free.ess_marker, self.op_free.masked_gather_map = 0, None, None, None
        self._dinv = None
        self._ws = None
        self._comm = None

    def use_stream(self, stream):
        """launch on `stream` (a torch stream) from now on -- e.g. a capturing stream for HIP-graph replay"""
        self.stream = C.c_void_p(stream.cuda_stream)

    # ---- BilinearForm::Mult (unconstrained L->L), the BK "Kernel" of bench_ceed.cpp:163-190
    def direct_active(self):
        """does mult() write the element-interior rows from the element kernel? (vector_cg.hip: constrained_mult)"""
        return bool(self.direct) and int(os.environ.get("MFEM_AMD_DIRECT", "1")) != 0

    def l2e(self, x, e0=0, e1=None, y=None):
        """element kernel over the elements [e0, e1) (e0 on a batch boundary of the blocked layout); with y, and where
        mult() does so, the element-interior rows go to y as in mult()"""
        e1 = self.ne if e1 is None else e1
        if e1 <= e0:
            return
        if y is not None and self.direct_active():
            # interior rows straight to y, everything else to the packed E-vector (mfem_amd_pa_apply_L2E_packed)
            D3, ni = self.D**3, (self.D - 2)**3
            nc = 6 if self.kind else 1
            pa_off = capi.lib.mfem_amd_pa_data_size(self.kind, 1, self.Q, e0) if self.layout else e0*nc*self.Q**3
            capi.call("mfem_amd_pa_apply_L2E_packed", self.kind, self.layout, self.D, self.Q, e1 - e0, self.B, self.G,
                      C.c_void_p(self.pa.data_ptr() + 8*pa_off), C.c_void_p(self.gm.data_ptr() + 4*e0*D3), _p(x),
                      C.c_void_p(self.yE.data_ptr() + 8*e0*(D3 - ni)),
                      C.c_void_p(y.data_ptr() + 8*(self.nscalar - self.direct + e0*ni)), self.stream)
            return
        D3, vd = self.D**3, max(1, self.vdim)
        nc = 6 if self.kind else 1
        pa_off = capi.lib.mfem_amd_pa_data_size(self.kind, 1, self.Q, e0) if self.layout else e0*nc*self.Q**3
        pa = C.c_void_p(self.pa.data_ptr() + 8*pa_off)
        gm = C.c_void_p(self.gm.data_ptr() + 4*e0*D3)
        yE = C.c_void_p(self.yE.data_ptr() + 8*e0*D3*vd)
        if self.vdim:
            capi.call("mfem_amd_pa_apply_L2E_vdim", self.kind, self.layout, self.D, self.Q, e1 - e0, self.vdim, 1, self.nscalar,
                      self.B, self.G, pa, 0, gm, None, _p(x), yE, self.stream)
            return
        capi.call("mfem_amd_pa_apply_L2E", self.kind, self.layout, self.D, self.Q, e1 - e0, self.B, self.G, pa,
                  gm, _p(x), yE, self.stream)

    def e2l(self, y):
        if self.vdim:
            capi.call("mfem_amd_restrict_scatter_vdim", self.nscalar, self.D**3, self.vdim, 1, _p(self.off), _p(self.ind),
                      _p(self.yE), _p(y), 0, 0, self.stream)
            return
        rows = self.ndofs - self.direct if self.direct_active() else self.ndofs        # as mult() does
        ind = self.pind if self.direct_active() else self.ind
        capi.call("mfem_amd_restrict_scatter", rows, _p(self.off), _p(ind), _p(self.yE), _p(y), 0, 0, self.stream)

    def mult(self, x, y):
        """y = A x; in parallel x's ghost entries are refreshed first and y ends with owners
        summed / ghosts zero (P^T A P)."""
        if self.ex is not None and not self.vdim:
            # P^T A P in the library (mfem_amd_pa_par_mult): the exchange callbacks are the transport's, the owner -> ghost
            # exchange runs on a second stream beside the interior elements (op.e_split)
            if self._comm is None:
                self._ws = torch.empty(capi.lib.mfem_amd_pcg_workspace_bytes(self.ndofs, 2000), dtype=torch.uint8, device=self.dev)
                self._comm = self.ex.make_comm(self._ws)
            rc = capi.lib.mfem_amd_pa_par_mult(C.byref(self.op_free), C.byref(self._comm), _p(x), _p(y), self.stream)
            if rc != 0:
                raise capi.MfemAmdError(f"{capi.lib.mfem_amd_last_error().decode()} / {getattr(self.ex, 'error', None)!r}")
            return
        if self.ex is None and not self.vdim:
            capi.call("mfem_amd_pa_apply_L2L", self.kind, self.layout, self.D, self.Q, self.ne, self.nscalar, self.B, self.G,
                      _p(self.pa), _p(self.gm), _p(self.off), _p(self.ind), _p(self.pind), self.direct, _p(x), _p(self.yE), _p(y),
                      self.stream)
            return
        if self.ex is not None:
            self.ex.forward(x.data_ptr(), self.stream)
        self.l2e(x, y=y)
        self.e2l(y)
        if self.ex is not None:
            self.ex.reverse(y.data_ptr(), self.stream)

    # ---- OperatorJacobiSmoother(a, ess) (linalg/solvers.cpp:331-348,401-425)
    def dinv(self):
        if self._dinv is None:
            dE = torch.zeros(self.ne*self.D**3, dtype=torch.float64, device=self.dev)
            diag = torch.empty(self.nscalar, dtype=torch.float64, device=self.dev)
            if self.kind:
                capi.call("mfem_amd_pa_diffusion_diag_3d", self.D, self.Q, self.ne, 1, self.B, self.G, _p(self.pa),
                          self.layout, _p(dE), self.stream)
            else:
                capi.call("mfem_amd_pa_mass_diag_3d", self.D, self.Q, self.ne, self.B, _p(self.pa), self.layout, _p(dE),
                          self.stream)
            capi.call("mfem_amd_restrict_scatter", self.nscalar, _p(self.off), _p(self.ind), _p(dE), _p(diag), 0, 1, self.stream)
            if self.vdim:           # scalar coefficient: every component carries the scalar diagonal
                dv = torch.empty(self.ndofs, dtype=torch.float64, device=self.dev)
                capi.call("mfem_amd_vec_replicate_vdim", self.nscalar, self.vdim, 1, None, _p(diag), _p(dv), self.stream)
                torch.cuda.current_stream(self.dev).synchronize()
                diag = dv
            if self.ex is not None:                 # ParBilinearForm::AssembleDiagonal: one P^T exchange
                self.ex.reverse(diag.data_ptr(), self.stream)
                if self.ghost.numel():              # keep 1/0 out of the ghost entries
                    capi.call("mfem_amd_set_subvector", self.ghost.numel(), _p(self.ghost), None, 1.0, _p(diag), self.stream)
            self.diag = diag
            dinv = torch.empty_like(diag)
            capi.call("mfem_amd_jacobi_setup", self.ndofs, _p(diag), 1.0, self.ess.numel(), _p(self.ess), _p(dinv), self.stream)
            if self.ex is not None and self.ghost.numel():
                capi.call("mfem_amd_set_subvector", self.ghost.numel(), _p(self.ghost), None, 0.0, _p(dinv), self.stream)
            self._dinv = dinv
        return self._dinv

    def rhs_unit_source(self):
        """b_i = (1, phi_i) with essential rows eliminated (x_ess = 0): b = M 1 since the basis
        is a partition of unity.  In parallel: owners summed, ghosts zero."""
        # mass data on THIS problem's rule (so BP5 uses its GLL rule)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)
        from .tables import gauss_legendre, gauss_lobatto, lagrange_tables
        Qm = self.Q
        qx, qw = (gauss_lobatto(Qm) if Qm == self.D else gauss_legendre(Qm))
        Bm, Gm = self.B, self.G
        Bn, Gn = lagrange_tables(gauss_lobatto(self.Dn)[0], qx)
        W = np.ascontiguousarray((qw[None, None, :]*qw[None, :, None]*qw[:, None, None]).reshape(-1))
        pam = torch.zeros(capi.lib.mfem_amd_pa_data_size(0, 0, Qm, self.ne), dtype=torch.float64, device=self.dev)
        one = torch.ones(1, dtype=torch.float64, device=self.dev)
        Wd, nodesE = t(W), t(self._nodes_E)      # keep the device tensors alive across the launch
        capi.call("mfem_amd_pa_setup_from_nodes_3d", 0, self.Dn, Qm, 1, self.ne, Bn, Gn, _p(Wd), _p(nodesE), _p(one),
                  1, 0, _p(pam), None, self.stream)
        ones = torch.ones(self.nscalar, dtype=torch.float64, device=self.dev)
        b = torch.empty_like(ones)
        capi.call("mfem_amd_pa_apply_L2E", 0, 0, self.D, Qm, self.ne, Bm, Gm, _p(pam), _p(self.gm), _p(ones), _p(self.yE), self.stream)
        capi.call("mfem_amd_restrict_scatter", self.nscalar, _p(self.off), _p(self.ind), _p(self.yE), _p(b), 0, 0, self.stream)
        if self.vdim:               # f = (1,1,1)/sqrt(3) (bench_ceed.cpp:54): the scalar form on every component
            bv = torch.empty(self.ndofs, dtype=torch.float64, device=self.dev)
            sc = np.full(self.vdim, 1.0/np.sqrt(float(self.vdim)))
            capi.call("mfem_amd_vec_replicate_vdim", self.nscalar, self.vdim, 1, sc.ctypes.data_as(C.c_void_p), _p(b), _p(bv), self.stream)
            torch.cuda.current_stream(self.dev).synchronize()
            b = bv
        if self.ex is not None:
            self.ex.reverse(b.data_ptr(), self.stream)
        capi.call("mfem_amd_set_subvector", self.ess.numel(), _p(self.ess), None, 0.0, _p(b), self.stream)
        torch.cuda.current_stream(self.dev).synchronize()   # temporaries above die here
        return b

    # ---- CGSolver / PCG
    def pcg(self, b, rel_tol, abs_tol=0.0, max_iter=2000, check_every=8, jacobi=True, want_hist=False):
        nbytes = capi.lib.mfem_amd_pcg_workspace_bytes(self.ndofs, max_iter)
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
            self._comm = self.ex.make_comm(self._ws) if self.ex is not None else None
        x = torch.empty(self.ndofs, dtype=torch.float64, device=self.dev)
        res = capi.PcgResult()
        hist = np.zeros(max_iter + 2) if want_hist else None
        hp = hist.ctypes.data_as(C.c_void_p) if want_hist else None
        dinv = self.dinv() if jacobi else None
        if self.ex is None:
            capi.call("mfem_amd_pcg_pa", C.byref(self.op), _p(dinv), _p(b), _p(x), rel_tol, abs_tol, max_iter, check_every,
                      _p(self._ws), C.byref(res), hp, self.stream)
        else:
            rc = capi.lib.mfem_amd_pcg_pa_par(C.byref(self.op), C.byref(self._comm), _p(dinv), _p(b), _p(x), rel_tol, abs_tol,
                                              max_iter, check_every, _p(self._ws), C.byref(res), hp, self.stream)
            if rc != 0:
                raise capi.MfemAmdError(f"{capi.lib.mfem_amd_last_error().decode()} / {self.ex.error!r}")
        return x, res, (hist[:res.history_count] if want_hist else None)
