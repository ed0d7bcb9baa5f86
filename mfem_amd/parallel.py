"""Multi-GPU host layer: box partition of a Cartesian hex mesh, shared-dof tables and the
exchange that plugs into the C ABI's ``mfem_amd_comm`` (include/mfem_amd.h).

Reference counterparts (semantics restated from the text; the parallel reference cannot be
built here -- no MPI/hypre/METIS):
  * ParMesh element partition: ``Mesh::CartesianPartitioning`` (mesh/mesh.cpp:8952), one box
    per rank, rank = rx + PX*(ry + PY*rz).
  * true-dof ownership: master of a dof's group = lowest rank among the sharers
    (general/communication.cpp:120-124, communication.hpp:195-203).
  * P / P^T: ``DeviceConformingProlongationOperator::Mult / MultTranspose``
    (fem/pfespace.cpp:5401-5447, 5503-5539): pack ``shr_buf[i] = x[shr_ltdof[i]]``, one message
    per neighbour each way, unpack ``y[ext_ldof[i]] = ext_buf[i]``; reverse direction with the
    deterministic CSR accumulate (``unq_ltdof / unq_shr_i / unq_shr_j``, :5475-5494).
  * global dot: local dot + all-reduce of ONE double (linalg/vector.hpp:773-779).

MI355X mapping: one process per GPU; the transport is ``torch.distributed`` -- backend
"nccl" is RCCL, so the grouped isend/irecv below become one ncclGroupStart/ncclSend/ncclRecv/
ncclGroupEnd per exchange and, with <= 7 face/edge/corner neighbours per direction on a
2x2x2 grid, every neighbour pair has its own xGMI link (the default transport issues them as one all_to_all_single with zero splits for non-neighbours; MFEM_AMD_EXCHANGE=p2p keeps the grouped isend/irecv).  Messages are tiny (a face of a
54^3 p=4 box is 47 089 doubles = 377 KB), so the exchange is latency-bound; the all-reduce
carries one double.  Vectors live on local L-dofs with ghost entries kept at zero, so no
T-vector <-> L-vector copies exist at all (the reference copies through ``ltdof_ldof``).
The pack/unpack/accumulate kernels are HIP (C ABI); with the "gloo" backend (CPU tests, or
several ranks sharing one GPU) buffers are staged through host memory.
"""
import ctypes as C
import itertools

import numpy as np
import torch
import torch.distributed as dist

from .cartesian import CartesianProblem


def process_grid(world):
    """Box process grid for `world` ranks (as cubic as possible, x fastest)."""
    best = None
    for px in range(1, world + 1):
        if world % px:
            continue
        for py in range(1, world//px + 1):
            if (world//px) % py:
                continue
            pz = world//(px*py)
            key = (max(px, py, pz) - min(px, py, pz), -px, -py)
            if best is None or key < best[0]:
                best = (key, (px, py, pz))
    return best[1]


def _finish_shared_tables(part, ndofs):
    """From part.upper / part.lower (lists of (neighbour rank, local dof ids), the ids in an order
    both sides agree on): the flat pack/unpack index arrays, the CSR of the deterministic accumulate
    (pfespace.cpp:5475-5494: unq_ltdof / unq_shr_i / unq_shr_j), the ghost mask and the owned count."""
    part.upper.sort(key=lambda t: t[0])
    part.lower.sort(key=lambda t: t[0])
    cat = lambda lst: (np.concatenate([a for _, a in lst]) if lst else np.zeros(0, np.int32)).astype(np.int32)
    part.shr_ltdof = cat(part.upper)          # owned dofs sent to sharers (repeats allowed)
    part.ext_ldof = cat(part.lower)           # ghost dofs (each appears exactly once)
    part.shr_sizes = [len(a) for _, a in part.upper]
    part.ext_sizes = [len(a) for _, a in part.lower]
    assert len(np.unique(part.ext_ldof)) == len(part.ext_ldof)
    # CSR for the deterministic accumulate: unique owned-shared dof -> positions in shr_buf
    order = np.argsort(part.shr_ltdof, kind="stable")
    uniq, counts = np.unique(part.shr_ltdof, return_counts=True)
    part.unq_ltdof = uniq.astype(np.int32)
    part.unq_shr_i = np.zeros(len(uniq) + 1, np.int32)
    np.cumsum(counts, out=part.unq_shr_i[1:])
    part.unq_shr_j = order.astype(np.int32)   # ascending buffer position = ascending neighbour rank
    part.ghost = np.zeros(ndofs, bool)
    part.ghost[part.ext_ldof] = True
    part.n_owned = int(ndofs - part.ghost.sum())


class BoxPartition:
    """Local problem and shared-dof tables of rank `rank` in a (PX,PY,PZ) box grid with
    n = (nx,ny,nz) elements per rank and H1 order p."""

    def __init__(self, p, n, pgrid, rank, extent=(1.0, 1.0, 1.0), numbering="lexicographic"):
        """numbering = "interior-last": the local elements are ordered interior first (those that touch no ghost dof:
        their part of an apply can run while the owner -> ghost exchange is in flight), and the local L-dofs are
        renumbered vertex/edge/face dofs first (in grid order), then the element-interior dofs element by element -- the
        class of numberings the reference gives every H1 space (fem/fespace.cpp) -- so that the element kernel writes
        those rows itself and the work E-vector is packed (include/mfem_amd.h: mfem_amd_interior_rows,
        mfem_amd_pack_indices), exactly as on one GPU.  self.perm[grid id] = local id (None for the grid numbering)."""
        self.p, self.n, self.pgrid, self.rank = p, tuple(n), tuple(pgrid), rank
        self.perm = None
        PX, PY, PZ = pgrid
        self.coord = (rank % PX, (rank//PX) % PY, rank//(PX*PY))
        gn = (PX*n[0], PY*n[1], PZ*n[2])
        origin = tuple(self.coord[d]*n[d] for d in range(3))
        self.prob = CartesianProblem(p, n[0], n[1], n[2], extent[0], extent[1], extent[2], origin=origin, global_n=gn)
        n1 = self.prob.n1
        has_lo = [self.coord[d] > 0 for d in range(3)]
        has_hi = [self.coord[d] < pgrid[d] - 1 for d in range(3)]
        self.has_lo, self.has_hi = has_lo, has_hi

        def index_set(off):
            """local L-dof ids of the set shared with the neighbour at offset `off`, in
            lexicographic (z slowest) order -- the same order on both sides"""
            rng = []
            for d in range(3):
                if off[d] == 1:
                    rng.append(np.array([n1[d] - 1]))
                elif off[d] == -1:
                    rng.append(np.array([0]))
                else:
                    rng.append(np.arange(1 if has_lo[d] else 0, n1[d]))
            ix, iy, iz = rng
            ids = ix[None, None, :] + n1[0]*(iy[None, :, None] + n1[1]*iz[:, None, None])
            return ids.reshape(-1).astype(np.int32)

        def nbr_rank(off):
            c = [self.coord[d] + off[d] for d in range(3)]
            return c[0] + PX*(c[1] + PY*c[2])

        # owner side ("shr", upper neighbours) and ghost side ("ext", lower neighbours),
        # neighbours sorted by rank (fixed accumulate order)
        self.upper, self.lower = [], []
        for off in itertools.product((0, 1), repeat=3):
            if any(off) and all(has_hi[d] for d in range(3) if off[d]):
                self.upper.append((nbr_rank(off), index_set(off)))
        for off in itertools.product((0, -1), repeat=3):
            if any(off) and all(has_lo[d] for d in range(3) if off[d]):
                self.lower.append((nbr_rank(off), index_set(off)))
        if numbering == "mfem":
            # the reference's numbering and element order for the local box (what ParFiniteElementSpace gives a rank:
            # its own vertex / edge / face / interior numbering of its own elements), tables from the C++ host layer
            from .hosttables import MfemBox
            self.prob = MfemBox(p, n[0], n[1], n[2], extent[0], extent[1], extent[2], origin=origin, global_n=gn)
            self.perm = self.prob.perm
            self.upper = [(r, self.perm[ids].astype(np.int32)) for r, ids in self.upper]
            self.lower = [(r, self.perm[ids].astype(np.int32)) for r, ids in self.lower]
        if numbering in ("interior-last", "mfem") and p >= 2:
            self._renumber_interior_last()
        _finish_shared_tables(self, self.prob.ndofs)

    def _renumber_interior_last(self):
        pr, D = self.prob, self.p + 1
        nd, ni = D**3, (D - 2)**3
        gm = np.asarray(pr.gather_map).reshape(pr.ne, nd)
        ghost = np.zeros(pr.ndofs, bool)
        for _, ids in self.lower:
            ghost[ids] = True
        iface = ghost[gm].any(axis=1)
        eperm = np.concatenate([np.flatnonzero(~iface), np.flatnonzero(iface)])     # interior elements first, stable
        gm = gm[eperm]
        pr.nodes_E = np.ascontiguousarray(np.asarray(pr.nodes_E).reshape(pr.ne, -1)[eperm].reshape(-1))
        inter = gm.reshape(pr.ne, D, D, D)[:, 1:-1, 1:-1, 1:-1].reshape(-1)          # (element, lexicographic interior index)
        is_int = np.zeros(pr.ndofs, bool)
        is_int[inter] = True
        i0 = pr.ndofs - pr.ne*ni
        perm = np.empty(pr.ndofs, np.int64)
        perm[~is_int] = np.arange(i0)
        perm[inter] = i0 + np.arange(pr.ne*ni)
        # (composed with a numbering applied before: self.perm maps grid ids all the way)
        self.perm = perm if self.perm is None else perm[self.perm]
        self.element_order = eperm
        pr.gather_map = np.ascontiguousarray(perm[gm.reshape(-1)].astype(np.int32))
        # E -> L table of the new order: per dof, ascending E-vector position (fem/restriction.cpp:86-99)
        pr.indices = np.argsort(pr.gather_map, kind="stable").astype(np.int32)
        pr.offsets = np.zeros(pr.ndofs + 1, np.int32)
        np.cumsum(np.bincount(pr.gather_map, minlength=pr.ndofs), out=pr.offsets[1:])
        # the shared sets keep their order (both sides list them in grid order), only the ids change
        self.upper = [(r, perm[ids].astype(np.int32)) for r, ids in self.upper]
        self.lower = [(r, perm[ids].astype(np.int32)) for r, ids in self.lower]

    def local_to_global(self):
        """global lexicographic dof id of every local L-dof (tests)."""
        p, n1 = self.p, self.prob.n1
        PX, PY, PZ = self.pgrid
        g1 = (p*PX*self.n[0] + 1, p*PY*self.n[1] + 1, p*PZ*self.n[2] + 1)
        o = [self.coord[d]*self.n[d]*p for d in range(3)]
        ix, iy, iz = np.arange(n1[0]) + o[0], np.arange(n1[1]) + o[1], np.arange(n1[2]) + o[2]
        l2g = (ix[None, None, :] + g1[0]*(iy[None, :, None] + g1[1]*iz[:, None, None])).reshape(-1)
        if self.perm is not None:
            out = np.empty_like(l2g)
            out[self.perm] = l2g
            l2g = out
        return l2g, g1

    def essential_marker(self):
        """1 = essential & owned, 2 = essential ghost copy, 0 otherwise; plus the dof list."""
        grid = getattr(self.prob, "_grid", self.prob)          # (boundary_dofs of the grid numbering, mapped below)
        ess = grid.boundary_dofs()
        if self.perm is not None:
            ess = np.sort(self.perm[ess]).astype(np.int32)
        m = np.zeros(self.prob.ndofs, np.uint8)
        m[ess] = 1
        m[ess[self.ghost[ess]]] = 2
        return m, ess


class MeshTables:
    """Global (serial) tables of an H1 space on a hex mesh, as every rank holds them before the
    partition (the reference builds ParMesh from the serial Mesh on every rank, mesh/pmesh.cpp:106-):
    lexicographic element->dof map, E-nodes of the mesh's nodal space, essential dofs."""

    def __init__(self, D1D, ndofs, gather_map, nodes_E, Dn, ess):
        self.D1D, self.ndofs, self.Dn = int(D1D), int(ndofs), int(Dn)
        nd = self.D1D**3
        self.gather_map = np.ascontiguousarray(gather_map, np.int32).reshape(-1, nd)
        self.ne = self.gather_map.shape[0]
        self.nodes_E = np.ascontiguousarray(nodes_E, np.float64).reshape(self.ne, 3*self.Dn**3)
        self.ess = np.ascontiguousarray(ess, np.int32)

    @classmethod
    def from_cartesian(cls, prob):
        return cls(prob.p + 1, prob.ndofs, prob.gather_map, prob.nodes_E, 2, prob.boundary_dofs())

    @classmethod
    def from_case(cls, c):
        """tests/golden case dict, or the files written by examples/bin/dump_tables (same keys)."""
        return cls(c["D1D"], c["ndofs"], c["gather_map"], c["nodes_E"], c["nodes_D1D"], c["ess_tdof"])


def chunk_partition(ne, world):
    """Element partition without METIS: contiguous chunks of the element order (MFEM's Cartesian
    meshes are Hilbert-ordered, so chunks are compact).  partition[e] = rank."""
    return (np.arange(ne, dtype=np.int64)*world//ne).astype(np.int32)


class _LocalProblem:
    pass


class GeneralPartition:
    """Rank `rank`'s share of an arbitrary element partition of an unstructured hex mesh
    (SURVEY 8f-3): local elements, local dof numbering, and the shared-dof tables the exchange
    needs.  Restates what ParFiniteElementSpace derives from the group topology
    (fem/pfespace.cpp:1804-1851, general/communication.cpp:97-267), from the serial tables:
      * a dof is shared by the ranks whose elements touch it (its group);
      * its owner (group master) is the lowest rank of the group (communication.cpp:120-124);
      * exchanges run owner <-> sharers, lists ordered by global dof id on both sides.
    Local dofs are numbered by ascending global id; local elements keep their global order.
    The partition itself (METIS in the reference, mesh/mesh.cpp:9106-9195) is an input here:
    any vector works, `chunk_partition` is the METIS-free default -- parity of the partition is
    unpinned, the partitioned solve is pinned to the serial one (tests/test_parallel_gloo.py)."""

    def __init__(self, tables, partition, rank):
        partition = np.asarray(partition)
        assert partition.shape == (tables.ne,)
        self.tables, self.rank = tables, rank
        nd = tables.D1D**3
        mine = np.flatnonzero(partition == rank)
        self.elements = mine
        gm = tables.gather_map[mine]                                  # (ne_loc, nd) global ids
        l2g = np.unique(gm)                                           # ascending global id
        self.l2g = l2g
        pr = _LocalProblem()
        pr.ne, pr.ndofs, pr.Dn = len(mine), len(l2g), tables.Dn
        pr.gather_map = np.searchsorted(l2g, gm).astype(np.int32).reshape(-1)
        order = np.argsort(pr.gather_map, kind="stable")              # element-major within a dof (restriction.cpp:86-99)
        pr.indices = order.astype(np.int32)
        pr.offsets = np.zeros(pr.ndofs + 1, np.int32)
        np.cumsum(np.bincount(pr.gather_map, minlength=pr.ndofs), out=pr.offsets[1:])
        pr.nodes_E = np.ascontiguousarray(tables.nodes_E[mine].reshape(-1))
        self.prob = pr
        # groups: (dof, rank) pairs over the whole mesh
        pairs = np.unique(np.stack([tables.gather_map.reshape(-1).astype(np.int64),
                                    np.repeat(partition.astype(np.int64), nd)], axis=1), axis=0)
        dofs, ranks = pairs[:, 0], pairs[:, 1]
        first = np.r_[True, dofs[1:] != dofs[:-1]]
        owner_of = np.full(tables.ndofs, -1, np.int64)
        owner_of[dofs[first]] = ranks[first]                          # pairs are sorted: first = lowest rank
        self.owner = owner_of[l2g]
        # dofs I own that other ranks touch: those ranks get my value (P) and send me their sums (P^T)
        mine_shared = (owner_of[dofs] == rank) & (ranks != rank)
        up, lo = {}, {}
        for g, r in zip(dofs[mine_shared], ranks[mine_shared]):
            up.setdefault(int(r), []).append(g)
        for g in l2g[self.owner != rank]:
            lo.setdefault(int(owner_of[g]), []).append(g)
        loc = lambda gl: np.searchsorted(l2g, np.sort(np.asarray(gl, np.int64))).astype(np.int32)
        self.upper = [(r, loc(g)) for r, g in up.items()]
        self.lower = [(r, loc(g)) for r, g in lo.items()]
        _finish_shared_tables(self, pr.ndofs)

    def local_to_global(self):
        return self.l2g, None

    def essential_marker(self):
        """1 = essential & owned, 2 = essential ghost copy, 0 otherwise; plus the local dof list."""
        ess = np.flatnonzero(np.isin(self.l2g, self.tables.ess)).astype(np.int32)
        m = np.zeros(self.prob.ndofs, np.uint8)
        m[ess] = 1
        m[ess[self.ghost[ess]]] = 2
        return m, ess


class _VdofView:
    """The shared-dof tables of a partition expanded from dofs to the vdofs of a byVDIM vector space."""

    def __init__(self, part, vd):
        ex = lambda ids: (np.asarray(ids, np.int64)[:, None]*vd + np.arange(vd)[None, :]).reshape(-1).astype(np.int32)
        self.upper = [(r, ex(ids)) for r, ids in part.upper]
        self.lower = [(r, ex(ids)) for r, ids in part.lower]
        self.prob = part.prob
        _finish_shared_tables(self, part.prob.ndofs*vd)


class Exchange:
    """P / P^T / all-reduce on top of torch.distributed.  `device` is a torch device; with
    CPU tensors (tests) the index kernels are torch ops, with CUDA tensors they are the HIP
    kernels of the C ABI called on raw pointers."""

    def __init__(self, part, device, group=None, vdim=0):
        """vdim > 0: the vectors are those of a vector space ordered byVDIM (vdof = dof*vdim + c): every
        shared dof carries vdim consecutive entries."""
        if vdim:
            part = _VdofView(part, vdim)
        self.part, self.device, self.group = part, torch.device(device), group
        self.on_gpu = self.device.type == "cuda"
        self.stage = self.on_gpu and dist.is_initialized() and dist.get_backend(group) == "gloo"
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
        self.shr_ltdof, self.ext_ldof = t(part.shr_ltdof), t(part.ext_ldof)
        self.unq_ltdof, self.unq_shr_i, self.unq_shr_j = t(part.unq_ltdof), t(part.unq_shr_i), t(part.unq_shr_j)
        self.shr_buf = torch.zeros(max(1, len(part.shr_ltdof)), dtype=torch.float64, device=self.device)
        self.ext_buf = torch.zeros(max(1, len(part.ext_ldof)), dtype=torch.float64, device=self.device)
        if self.stage:
            self.shr_host = torch.zeros_like(self.shr_buf, device="cpu").pin_memory()
            self.ext_host = torch.zeros_like(self.ext_buf, device="cpu").pin_memory()
        if self.on_gpu:
            from . import capi
            self.capi = capi
        self._cb_keep = None
        self._sum_views = {}
        self._ops = {}
        import os
        self.use_alltoall = os.environ.get("MFEM_AMD_EXCHANGE", "alltoall") != "p2p"
        # a peer may appear once per direction only (true for box and general partitions: one list per neighbour)
        for lst in (part.upper, part.lower):
            peers = [pr for pr, _ in lst]
            if len(set(peers)) != len(peers):
                self.use_alltoall = False

    # -- index kernels ---------------------------------------------------------------
    def _ptr(self, v):
        return v if isinstance(v, int) else v.data_ptr()

    def _pack(self, idx, v, buf, stream):
        if self.on_gpu:
            self.capi.call("mfem_amd_shared_pack", idx.numel(), C.c_void_p(idx.data_ptr()), C.c_void_p(self._ptr(v)),
                           C.c_void_p(buf.data_ptr()), stream)
        else:
            buf[:idx.numel()] = v[idx.long()]

    def _unpack(self, idx, buf, v, stream):
        if self.on_gpu:
            self.capi.call("mfem_amd_shared_unpack", idx.numel(), C.c_void_p(idx.data_ptr()), C.c_void_p(buf.data_ptr()),
                           C.c_void_p(self._ptr(v)), stream)
        else:
            v[idx.long()] = buf[:idx.numel()]

    def _accumulate(self, v, stream):
        nu = self.unq_ltdof.numel()
        if nu == 0:
            return
        if self.on_gpu:
            self.capi.call("mfem_amd_shared_accumulate", nu, C.c_void_p(self.unq_ltdof.data_ptr()),
                           C.c_void_p(self.unq_shr_i.data_ptr()), C.c_void_p(self.unq_shr_j.data_ptr()),
                           C.c_void_p(self.shr_buf.data_ptr()), C.c_void_p(self._ptr(v)), stream)
        else:
            off = self.unq_shr_i.tolist()
            for i, k in enumerate(self.unq_ltdof.tolist()):     # fixed order, small (tests)
                s = v[k]
                for j in range(off[i], off[i + 1]):
                    s = s + self.shr_buf[self.unq_shr_j[j]]
                v[k] = s

    def _zero_ghosts(self, v, stream):
        n = self.ext_ldof.numel()
        if n == 0:
            return
        if self.on_gpu:
            self.capi.call("mfem_amd_set_subvector", n, C.c_void_p(self.ext_ldof.data_ptr()), None, 0.0,
                           C.c_void_p(self._ptr(v)), stream)
        else:
            v[self.ext_ldof.long()] = 0.0

    # -- transport -------------------------------------------------------------------
    def _stream_ctx(self, stream):
        """Everything the transport issues (collectives, staging copies) is ordered on the hipStream_t the caller
        handed over -- the contract of mfem_amd_comm (include/mfem_amd.h): `stream` may differ from torch's
        current stream.  None / the null pointer = the stream torch is on."""
        import contextlib
        ptr = stream.value if isinstance(stream, C.c_void_p) else stream
        if not self.on_gpu or not ptr:
            return contextlib.nullcontext()
        return torch.cuda.stream(torch.cuda.ExternalStream(int(ptr), device=self.device))

    def _transfer(self, send_buf, send_list, send_sizes, recv_buf, recv_list, recv_sizes, send_host=None, recv_host=None):
        alone = not dist.is_initialized() or dist.get_world_size(self.group) == 1
        if not send_list and not recv_list and (alone or not self.use_alltoall):
            return          # point-to-point: nothing to post.  The all-to-all is a collective: every rank calls it,
                            # with all-zero splits if it has no neighbour (a rank that skipped it would hang the others)
        if self.stage:
            torch.cuda.current_stream().synchronize()
            send_host.copy_(send_buf)
            sb, rb = send_host, recv_host
        else:
            sb, rb = send_buf, recv_buf
        if self.use_alltoall:
            # ONE collective call per exchange: an all-to-all with per-rank split sizes that are zero for
            # non-neighbours (RCCL issues a grouped ncclSend/ncclRecv for the non-zero ones only).  The
            # neighbour lists are sorted by rank, so the buffers already are in rank order.  Host time per
            # exchange matters: the messages are tiny and the device work of an apply is ~0.45 ms.
            key = ("a2a", sb.data_ptr(), rb.data_ptr())
            sp = self._ops.get(key)
            if sp is None:
                world = dist.get_world_size(self.group)
                ins, outs = [0]*world, [0]*world
                for (peer, _), sz in zip(send_list, send_sizes):
                    ins[peer] += sz
                for (peer, _), sz in zip(recv_list, recv_sizes):
                    outs[peer] += sz
                sp = (ins, outs, sum(ins), sum(outs))
                self._ops[key] = sp
            ins, outs, ns, nr = sp
            dist.all_to_all_single(rb[:nr], sb[:ns], outs, ins, group=self.group)
            if self.stage:
                recv_buf.copy_(recv_host)
            return
        # point-to-point variant (MFEM_AMD_EXCHANGE=p2p): the buffers are persistent, so the P2POp list of a
        # direction is built once
        key = (sb.data_ptr(), rb.data_ptr())
        ops = self._ops.get(key)
        if ops is None:
            ops, o = [], 0
            for (peer, _), sz in zip(recv_list, recv_sizes):
                ops.append(dist.P2POp(dist.irecv, rb[o:o + sz], peer, self.group)); o += sz
            o = 0
            for (peer, _), sz in zip(send_list, send_sizes):
                ops.append(dist.P2POp(dist.isend, sb[o:o + sz], peer, self.group)); o += sz
            self._ops[key] = ops
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        if self.stage:
            recv_buf.copy_(recv_host)

    def forward(self, v, stream=None):
        """P: ghost entries of v := owner values.  v: tensor or raw device pointer."""
        p = self.part
        self._pack(self.shr_ltdof, v, self.shr_buf, stream)
        with self._stream_ctx(stream):
            self._transfer(self.shr_buf, p.upper, p.shr_sizes, self.ext_buf, p.lower, p.ext_sizes,
                           getattr(self, "shr_host", None), getattr(self, "ext_host", None))
        self._unpack(self.ext_ldof, self.ext_buf, v, stream)

    def reverse(self, v, stream=None):
        """P^T: owners += ghost contributions (fixed neighbour order); ghosts := 0."""
        p = self.part
        self._pack(self.ext_ldof, v, self.ext_buf, stream)
        with self._stream_ctx(stream):
            self._transfer(self.ext_buf, p.lower, p.ext_sizes, self.shr_buf, p.upper, p.shr_sizes,
                           getattr(self, "ext_host", None), getattr(self, "shr_host", None))
        self._accumulate(v, stream)
        self._zero_ghosts(v, stream)

    def allreduce_sum(self, t, stream=None):
        with self._stream_ctx(stream):
            if self.stage:
                h = t.cpu()                          # synchronises the (now current) stream
                dist.all_reduce(h, group=self.group)
                t.copy_(h)
            else:
                dist.all_reduce(t, group=self.group)

    # -- the C ABI communicator ---------------------------------------------------------
    def make_comm(self, workspace):
        """struct mfem_amd_comm whose callbacks run this exchange.  `workspace` is the torch
        uint8 tensor handed to mfem_amd_pcg_pa_par (the all-reduce slot lives inside it)."""
        from . import capi
        AR, EX = capi.AR_FN, capi.EX_FN
        self.error = None

        def guard(fn):
            def w(*a):
                try:
                    fn(*a)
                    return 0
                except Exception as e:                      # noqa: BLE001 -- must not unwind into C
                    self.error = e
                    return 1
            return w

        self._sum_views = {}       # views into THIS workspace only (an earlier, re-allocated one must never be reduced)

        def allreduce(ctx, buf, n, stream):
            off = buf - workspace.data_ptr()
            key = (off, n)
            if key not in self._sum_views:
                self._sum_views[key] = workspace[off:off + 8*n].view(torch.float64)
            self.allreduce_sum(self._sum_views[key], C.c_void_p(stream))

        cbs = (AR(guard(allreduce)), EX(guard(lambda ctx, v, s: self.forward(v, C.c_void_p(s)))),
               EX(guard(lambda ctx, v, s: self.reverse(v, C.c_void_p(s)))))
        self._cb_keep = cbs
        return capi.Comm(None, cbs[0], cbs[1], cbs[2])
