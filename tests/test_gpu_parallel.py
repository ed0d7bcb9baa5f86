"""-m gpu: the partitioned solve through the C ABI (mfem_amd_pcg_pa_par + the exchange of
mfem_amd/parallel.py), 2 ranks sharing cuda:0 over gloo (host-staged messages), against the
serial device solve of the same global problem AND against the CPU oracle on the global problem:
identical iteration count, solution to 1e-10, operator action to 1e-12."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _oracle_checks(problem, p, sp, bp, l2g, own, xg, yl, X, b):
    """The partitioned device results of this rank against the CPU ORACLE on the global problem (not only against the
    serial device solve), for every kind of partition (boxes, random element assignments, the curved toroid tables) and
    both transports: operator action to 1e-12, Jacobi-PCG iteration count EQUAL where it is decided outside the
    rounding floor, solutions to 1e-9 at the tight tolerance."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib as orc
    from mfem_amd.tables import bakeoff_tables
    kind, D, Q, B, G, _, _, W = bakeoff_tables(problem, p)
    gp, gne = sp.prob, sp.ne
    Dn = getattr(gp, "Dn", 2)
    qx, _ = orc.gauss_lobatto(Q) if problem in ("bp5", "bp6") else orc.gauss_legendre(Q)
    Bn, Gn = orc.basis_1d(Dn - 1, qx)
    J = np.zeros(9*Q**3*gne)
    orc.lib.orc_jacobians_3d(Dn, Q, gne, Bn, Gn, gp.nodes_E, J)
    det = np.zeros(Q**3*gne); orc.lib.orc_det_3d(Q, gne, J, det)
    if kind:
        pa = np.zeros(6*Q**3*gne); orc.lib.orc_diffusion_setup_3d(Q, 1, gne, W, J, np.ones(1), 1, pa)
    else:
        pa = np.zeros(Q**3*gne); orc.lib.orc_mass_setup_3d(Q, gne, W, det, np.ones(1), 1, pa)
    goff, gind = sp.off.cpu().numpy(), sp.ind.cpu().numpy()
    gmap = sp.gm.cpu().numpy()
    gop = orc.Operator(kind, D, Q, gne, sp.nscalar, B, G, pa, gmap, goff, gind, sp.ess_host)
    yo = orc.Operator(kind, D, Q, gne, sp.nscalar, B, G, pa, gmap, goff, gind).mult(xg.cpu().numpy())
    e3 = np.max(np.abs(yl.cpu().numpy()[own] - yo[l2g][own]))/np.max(np.abs(yo))
    assert e3 < 1e-12, e3
    bE = np.zeros(gne*D**3); orc.lib.orc_domain_lf_3d(D, Q, gne, B, W, det, 1.0, bE)
    bo = np.zeros(sp.nscalar); orc.lib.orc_restriction_mult_transpose(sp.nscalar, goff, gind, bE, bo, 0, 0)
    bo[sp.ess_host] = 0.0
    # iteration count at ex1's stopping rule (rel tol 1e-6 on sqrt((B r, r)), examples/ex1.cpp:251), where the CG
    # recurrences of the two implementations still agree to rounding: EQUAL.  Far below that the residual histories of
    # a fast-converging solve drift apart by O(1) factors (loss of orthogonality amplifies the dot-product order), so
    # at 1e-10 the SOLUTIONS are compared, not the counts.
    dinv_o = orc.jacobi_setup(gop.diag(), sp.ess_host)
    Xo6, its_o6, ho6, conv_o6 = gop.pcg(bo, dinv_o, 1e-6, max_iter=1000)
    X6, res6, hist6 = bp.pcg(b, 1e-6, max_iter=1000, check_every=4, want_hist=True)
    assert res6.iterations == its_o6 and conv_o6 and res6.converged == 1, (res6.iterations, its_o6, hist6[-3:], ho6[-3:])
    assert np.max(np.abs(hist6 - ho6)/ho6) < 1e-6, np.max(np.abs(hist6 - ho6)/ho6)
    Xo, its_o, ho, conv_o = gop.pcg(bo, dinv_o, 1e-10, max_iter=1000)
    assert conv_o
    eo = np.max(np.abs(X.cpu().numpy() - Xo[l2g]))/np.max(np.abs(Xo))
    assert eo < 1e-9, eo


def _worker(rank, world, port, problem, p, n, out, mode="box"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, str(ROOT))
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mfem_amd import capi
        from mfem_amd.bp import BakeoffProblem
        from mfem_amd.parallel import BoxPartition, Exchange, GeneralPartition, MeshTables, process_grid
        capi.call("mfem_amd_init", 0)
        torch.cuda.set_device(0)
        pg = process_grid(world)
        if mode == "box-mfem":
            # what bench.py --gpus N runs: the reference's numbering of each rank's box, interior elements first, the
            # element-interior rows written by the element kernel and the packed E-vector -- also inside the solver
            os.environ["MFEM_AMD_DIRECT"] = "2"
            part = BoxPartition(p, (n, n, n), pg, rank, numbering="mfem")
        elif mode == "box":
            part = BoxPartition(p, (n, n, n), pg, rank)
        else:
            # arbitrary element partition of a Cartesian mesh's tables, or of a curved mesh file's
            # tables taken from the reference fixtures (order-3 nodes)
            sys.path.insert(0, str(ROOT / "tests"))
            from conftest import load_case
            from mfem_amd.cartesian import CartesianProblem
            tabs = (MeshTables.from_case(load_case("toroid-hex_diff_p3")) if mode == "toroid"
                    else MeshTables.from_cartesian(CartesianProblem(p, 2*n, n, n + 1)))
            pv = np.random.default_rng(5).integers(0, world, tabs.ne).astype(np.int32)
            part = GeneralPartition(tabs, pv, rank)
        vd = 3 if problem in ("bp2", "bp4", "bp6") else 1          # vector problems: vdofs, byVDIM
        ex = Exchange(part, "cuda:0", vdim=vd if vd > 1 else 0)
        bp = BakeoffProblem(problem, p, (n, n, n), "cuda:0", part=part, exchange=ex)
        assert mode != "box-mfem" or (bp.direct == bp.ne*(p - 1)**3 and bp.pind is not None), (bp.direct, bp.e_split)
        assert mode != "box-mfem" or not part.ghost.any() or bp.e_split > 0 or bp.ne < 14
        b = bp.rhs_unit_source()
        X, res, hist = bp.pcg(b, 1e-10, max_iter=1000, check_every=4, want_hist=True)
        # serial solve of the global problem on the same GPU
        if mode.startswith("box"):
            sp = BakeoffProblem(problem, p, (pg[0]*n, pg[1]*n, pg[2]*n), "cuda:0")
        else:
            sp = BakeoffProblem(problem, p, None, "cuda:0", part=GeneralPartition(tabs, np.zeros(tabs.ne, np.int32), 0))
        Xs, rs, hs = sp.pcg(sp.rhs_unit_source(), 1e-10, max_iter=1000, check_every=4, want_hist=True)
        torch.cuda.synchronize()
        l2g, _ = part.local_to_global()
        ghost_mask = part.ghost
        if vd > 1:
            l2g = (np.asarray(l2g, np.int64)[:, None]*vd + np.arange(vd)[None, :]).reshape(-1)
            ghost_mask = np.repeat(part.ghost, vd)
        assert res.iterations == rs.iterations, (res.iterations, rs.iterations)
        Xs_l = Xs.cpu().numpy()[l2g]
        err = np.max(np.abs(X.cpu().numpy() - Xs_l))/np.max(np.abs(Xs_l))
        assert err < 1e-10, err                     # owners AND ghosts consistent
        big = hs > 1e-12*hs[0]              # below that the history is dot-order rounding noise
        assert np.max(np.abs(hist[big] - hs[big])/hs[big]) < 1e-6
        # operator apply: P^T A P x vs serial
        xg = torch.rand(sp.ndofs, dtype=torch.float64, generator=torch.Generator().manual_seed(1)).cuda()
        ys = torch.empty_like(xg); sp.mult(xg, ys)
        xl = xg[torch.from_numpy(l2g).cuda()].clone()
        xl[bp.ghost.long()] = 0.0
        yl = torch.empty_like(xl); bp.mult(xl, yl)
        torch.cuda.synchronize()
        own = ~ghost_mask
        e2 = np.max(np.abs(yl.cpu().numpy()[own] - ys.cpu().numpy()[l2g][own]))/float(ys.abs().max())
        assert e2 < 1e-12, e2
        if vd == 1:
            _oracle_checks(problem, p, sp, bp, l2g, own, xg, yl, X, b)
        if mode.startswith("box") and vd == 1:
            nX = float(np.sqrt(np.sum(X.cpu().numpy()[own]**2)))
            # ---- the all-reduce slot follows the workspace: small, large, small max_iter (the workspace is
            # re-allocated when max_iter grows and the history block moves the slot)
            for mi in (10, 40, 10):
                Xa, ra, _ = bp.pcg(b, 0.0, max_iter=mi, check_every=5)
                Xb, rb, _ = sp.pcg(sp.rhs_unit_source(), 0.0, max_iter=mi, check_every=5)
                torch.cuda.synchronize()
                assert ra.iterations == rb.iterations == mi
                ea = np.max(np.abs(Xa.cpu().numpy() - Xb.cpu().numpy()[l2g]))/float(Xb.abs().max())
                assert ea < 1e-10, (mi, ea)
                if rb.final_nom > 1e-16*rb.initial_nom:          # below that (B r, r) is rounding noise
                    assert abs(ra.final_nom - rb.final_nom) <= 1e-6*abs(rb.final_nom), (mi, ra.final_nom, rb.final_nom)
            # ---- stream contract of the comm callbacks: the exchange on a stream that is NOT torch's current one
            import ctypes as C
            side = torch.cuda.Stream()
            with torch.cuda.stream(side):
                big = torch.rand(1 << 24, dtype=torch.float64, device="cuda:0")
                for _ in range(4):
                    big = big*1.0000001 + 1e-9                      # keeps `side` busy ahead of the fill below
                v = xg[torch.from_numpy(l2g).cuda()].clone()
                v[bp.ghost.long()] = -777.0
            ex.forward(v.data_ptr(), C.c_void_p(side.cuda_stream))   # current stream: the default one
            side.synchronize()
            assert torch.equal(v.cpu(), xg.cpu()[torch.from_numpy(l2g)])
        out.put((rank, "ok"))
    except Exception:                                                        # noqa: BLE001
        import traceback
        out.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("problem,p,n,mode", [("bp3", 3, 4, "box"), ("bp1", 2, 5, "box"), ("bp3", 2, 3, "random"),
                                              ("bp3", 3, 0, "toroid"), ("bp4", 2, 3, "box"), ("bp2", 3, 2, "random"),
                                              ("bp3", 4, 3, "box-mfem"), ("bp1", 3, 4, "box-mfem"), ("bp5", 5, 2, "box-mfem")])
def test_two_ranks_match_serial(problem, p, n, mode):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29700 + (os.getpid() % 1000) + 7*len(mode)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, problem, p, n, out, mode)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = [out.get(timeout=110) for _ in range(2)]
    for pr in procs:
        pr.join(30)
    bad = [m for _, m in res if m != "ok"]
    assert not bad, bad[0]


def test_ex1p_driver_two_ranks_equals_one_rank_and_serial_ex1():
    """examples/ex1p.py (one process per GPU; here 2 ranks over gloo on one GPU) on a curved mesh file: the same
    unknown count and (B r, r) history as one rank and as the serial C++ ex1 on the same refined mesh."""
    import re
    import subprocess
    mesh = str(ROOT / "tests" / "golden" / "meshes" / "toroid-hex.mesh")
    env = dict(os.environ, MFEM_AMD_DIST_BACKEND="gloo")

    def parse(txt):
        n = int(re.search(r"Number of finite element unknowns: (\d+)", txt).group(1))
        hist = [float(m) for m in re.findall(r"Iteration :\s+\d+\s+\(B r, r\) = (\S+)", txt)]
        return n, hist

    outs = {}
    for nr in (1, 2):
        port = 29850 + (os.getpid() % 500) + nr
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nr}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), str(ROOT / "examples" / "ex1p.py"), "-m", mesh, "-o", "2", "-r", "2"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=280, env=env, cwd=str(ROOT))
        assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-1500:]
        outs[nr] = parse(r.stdout)
        assert f"ranks = {nr}" in r.stdout
    r = subprocess.run([str(ROOT / "examples" / "bin" / "ex1"), "-m", mesh, "-o", "2", "-r", "2", "-pa", "-d", "hip", "-no-vis", "-no-save"],
                       capture_output=True, text=True, timeout=280, cwd=str(ROOT))
    assert r.returncode == 0, r.stderr[-1500:]
    n_ser, h_ser = parse(r.stdout)
    (n1, h1), (n2, h2) = outs[1], outs[2]
    assert n1 == n2 == n_ser
    # ex1p runs to rel tol 1e-12 (a 1e-24 reduction of (B r, r)): where the last decision sits in rounding.  The count is
    # compared where it is decided outside the rounding floor -- ex1's own stopping rule, rel tol 1e-6: the first iteration
    # with (B r, r) <= 1e-12 (B r, r)_0 is the SAME in all three runs
    first = lambda h: int(np.argmax(np.asarray(h) <= 1e-12*h[0]))
    assert first(h1) == first(h2) == len(h_ser) - 1 > 0, (first(h1), first(h2), len(h_ser) - 1)
    m = min(len(h1), len(h2), len(h_ser))              # ex1 stops earlier (PCG(...,1e-12) = rel tol 1e-6)
    h1, h2, h_ser = np.array(h1[:m]), np.array(h2[:m]), np.array(h_ser[:m])
    big = h_ser > 1e-10*h_ser[0]
    # same recurrence, different dof numbering and dot-product order: the rounding differences grow along CG
    # (2e-5 relative ten orders below the start); the first iterations agree to the 6 printed digits
    assert np.all(np.abs(h1[big] - h_ser[big]) <= 1e-3*h_ser[big]) and np.all(np.abs(h2[big] - h_ser[big]) <= 1e-3*h_ser[big])
    assert np.all(np.abs(h1[:6] - h_ser[:6]) <= 2e-6*h_ser[:6]) and np.all(np.abs(h2[:6] - h_ser[:6]) <= 2e-6*h_ser[:6])


_NCCL_ONE = r"""
import os, sys
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.environ["MFEM_AMD_ROOT"])
from mfem_amd import capi
from mfem_amd.bp import BakeoffProblem
from mfem_amd.parallel import BoxPartition, Exchange
torch.cuda.set_device(0)
capi.call("mfem_amd_init", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
part = BoxPartition(3, (6, 6, 6), (1, 1, 1), 0)
ex = Exchange(part, "cuda:0")
bp = BakeoffProblem("bp3", 3, (6, 6, 6), "cuda:0", part=part, exchange=ex)
X, res, hist = bp.pcg(bp.rhs_unit_source(), 1e-10, max_iter=500, check_every=4, want_hist=True)
sp = BakeoffProblem("bp3", 3, (6, 6, 6), "cuda:0")
Xs, rs, hs = sp.pcg(sp.rhs_unit_source(), 1e-10, max_iter=500, check_every=4, want_hist=True)
t = torch.ones(1, dtype=torch.float64, device="cuda:0"); dist.all_reduce(t); dist.barrier()
# the two transports of Exchange._transfer on RCCL, with the only peer a world of one has (itself): the all-to-all
# with explicit split sizes on a side stream (as the overlapped apply issues it), and the grouped isend/irecv
src = torch.arange(5, dtype=torch.float64, device="cuda:0"); dst = torch.zeros(8, dtype=torch.float64, device="cuda:0")
side = torch.cuda.Stream()
ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream())
with torch.cuda.stream(side):
    side.wait_event(ev)
    dist.all_to_all_single(dst[:5], src[:5], [5], [5])
    dist.all_to_all_single(dst[5:5], src[:0], [0], [0])
torch.cuda.current_stream().wait_stream(side)
dst2 = torch.zeros(5, dtype=torch.float64, device="cuda:0")
for r in dist.batch_isend_irecv([dist.P2POp(dist.irecv, dst2, 0), dist.P2POp(dist.isend, src, 0)]):
    r.wait()
torch.cuda.synchronize()
assert torch.equal(dst[:5], src) and torch.equal(dst2, src), (dst, dst2)
assert res.iterations == rs.iterations and res.converged == 1, (res.iterations, rs.iterations)
assert float((X - Xs).abs().max()) <= 1e-12*float(Xs.abs().max())
dist.destroy_process_group()
print("nccl-one ok", res.iterations)
"""


def test_rccl_backend_world_of_one():
    """The RCCL ("nccl") transport on the real device with a world of one rank: communicator creation, the
    device-side all-reduce of the PCG scalars issued from the C callbacks, stream ordering.  (Eight-GPU runs are
    the driver's; two ranks cannot share one GPU under RCCL -- the gloo tests above cover the exchange itself.)"""
    import subprocess
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29900 + os.getpid() % 90), MFEM_AMD_ROOT=str(ROOT),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _NCCL_ONE], capture_output=True, text=True, timeout=280, env=env, cwd=str(ROOT))
    assert r.returncode == 0 and "nccl-one ok" in r.stdout, r.stdout[-1500:] + r.stderr[-2500:]


# ---- the C++ transport (libmfem_amd_comm.so): no torch.distributed, no Python inside the PCG iteration ----------
def _native_worker(rank, world, port, problem, p, n, out, mode, transport):
    sys.path.insert(0, str(ROOT))
    try:
        import torch
        from mfem_amd import capi
        from mfem_amd.bp import BakeoffProblem
        from mfem_amd.comm import NativeExchange, World
        from mfem_amd.parallel import BoxPartition, GeneralPartition, MeshTables, process_grid
        torch.cuda.set_device(0)
        capi.call("mfem_amd_init", 0)
        w = World(rank, world, "127.0.0.1", port, 0, transport)
        assert (w.rank, w.size, w.transport) == (rank, world, transport)
        pg = process_grid(world)
        if mode == "box-mfem":
            os.environ["MFEM_AMD_DIRECT"] = "2"
            part = BoxPartition(p, (n, n, n), pg, rank, numbering="mfem")
        elif mode == "box":
            part = BoxPartition(p, (n, n, n), pg, rank)
        else:
            from mfem_amd.cartesian import CartesianProblem
            tabs = MeshTables.from_cartesian(CartesianProblem(p, 2*n, n, n + 1))
            pv = np.random.default_rng(5).integers(0, world, tabs.ne).astype(np.int32)
            part = GeneralPartition(tabs, pv, rank)
        vd = 3 if problem in ("bp2", "bp4", "bp6") else 1
        ex = NativeExchange(w, part, "cuda:0", vdim=vd if vd > 1 else 0)
        bp = BakeoffProblem(problem, p, (n, n, n), "cuda:0", part=part, exchange=ex)
        n_global = int(w.allreduce_host([part.n_owned*vd])[0])
        b = bp.rhs_unit_source()
        X, res, hist = bp.pcg(b, 1e-10, max_iter=1000, check_every=4, want_hist=True)
        if mode.startswith("box"):
            sp = BakeoffProblem(problem, p, (pg[0]*n, pg[1]*n, pg[2]*n), "cuda:0")
        else:
            sp = BakeoffProblem(problem, p, None, "cuda:0", part=GeneralPartition(tabs, np.zeros(tabs.ne, np.int32), 0))
        assert n_global == sp.ndofs, (n_global, sp.ndofs)
        Xs, rs, hs = sp.pcg(sp.rhs_unit_source(), 1e-10, max_iter=1000, check_every=4, want_hist=True)
        torch.cuda.synchronize()
        l2g, _ = part.local_to_global()
        ghost_mask = part.ghost
        if vd > 1:
            l2g = (np.asarray(l2g, np.int64)[:, None]*vd + np.arange(vd)[None, :]).reshape(-1)
            ghost_mask = np.repeat(part.ghost, vd)
        assert res.iterations == rs.iterations and res.converged == 1, (res.iterations, rs.iterations)
        Xs_l = Xs.cpu().numpy()[l2g]
        assert np.max(np.abs(X.cpu().numpy() - Xs_l))/np.max(np.abs(Xs_l)) < 1e-10
        big = hs > 1e-12*hs[0]
        assert np.max(np.abs(hist[big] - hs[big])/hs[big]) < 1e-6
        xg = torch.rand(sp.ndofs, dtype=torch.float64, generator=torch.Generator().manual_seed(1)).cuda()
        ys = torch.empty_like(xg); sp.mult(xg, ys)
        xl = xg[torch.from_numpy(l2g).cuda()].clone()
        xl[bp.ghost.long()] = 0.0
        yl = torch.empty_like(xl); bp.mult(xl, yl)
        # the callbacks honour a stream that is not torch's current one
        side = torch.cuda.Stream()
        with torch.cuda.stream(side):
            v = xg[torch.from_numpy(l2g).cuda()].clone()
            v[bp.ghost.long()] = -777.0
        import ctypes as C
        ex.forward(v.data_ptr(), C.c_void_p(side.cuda_stream))
        side.synchronize()
        torch.cuda.synchronize()
        assert torch.equal(v.cpu(), xg.cpu()[torch.from_numpy(l2g)])
        own = ~ghost_mask
        e2 = np.max(np.abs(yl.cpu().numpy()[own] - ys.cpu().numpy()[l2g][own]))/float(ys.abs().max())
        assert e2 < 1e-12, e2
        assert float(yl[bp.ghost.long()].abs().max()) == 0.0 if bp.ghost.numel() else True
        if vd == 1:
            _oracle_checks(problem, p, sp, bp, l2g, own, xg, yl, X, b)       # the transport that ships, against the oracle
            # the ghost -> owner exchange beside the transpose-gather (op.reverse_overlap: ghost rows summed first, the two
            # halves of P^T on two streams) ran in everything above; with it switched off -- the exchange after the
            # transpose-gather, one stream -- the same bits come out of the apply and of the solver
            assert bp.op.reverse_overlap == 1 and bp.op.n_ghost_rows == bp.ghost.numel()
            assert bool(ex.comm.reverse_begin) and bool(ex.comm.reverse_end)
            bp.op.reverse_overlap = bp.op_free.reverse_overlap = 0
            yl2 = torch.empty_like(xl); bp.mult(xl, yl2)
            X2, res2, hist2 = bp.pcg(b, 1e-10, max_iter=1000, check_every=4, want_hist=True)
            torch.cuda.synchronize()
            assert torch.equal(yl, yl2) and torch.equal(X, X2) and res2.iterations == res.iterations
            assert np.array_equal(hist, hist2)
            bp.op.reverse_overlap = bp.op_free.reverse_overlap = 1
        w.barrier()
        ex.close(); w.close()
        out.put((rank, "ok"))
    except Exception:                                                        # noqa: BLE001
        import traceback
        out.put((rank, traceback.format_exc()))


def _run_native(world, problem, p, n, mode, transport):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 30100 + (os.getpid() % 1000) + 11*len(mode) + 3*world
    procs = [ctx.Process(target=_native_worker, args=(r, world, port, problem, p, n, out, mode, transport)) for r in range(world)]
    for pr in procs:
        pr.start()
    res = [out.get(timeout=150) for _ in range(world)]
    for pr in procs:
        pr.join(30)
    bad = [m for _, m in res if m != "ok"]
    assert not bad, bad[0]


@pytest.mark.parametrize("world,problem,p,n,mode", [(2, "bp3", 3, 4, "box"), (2, "bp1", 2, 5, "box"), (3, "bp3", 2, 3, "random"),
                                                    (4, "bp3", 2, 3, "box"), (2, "bp4", 2, 3, "box"), (2, "bp3", 4, 3, "box-mfem"),
                                                    (4, "bp3", 3, 3, "box-mfem")])
def test_cxx_transport_ranks_on_one_gpu_match_serial(world, problem, p, n, mode):
    """libmfem_amd_comm.so with the host-staged TCP transport (RCCL refuses two ranks on one device): the exchange
    tables, message order, deterministic accumulate and all-reduce of the C++ communicator, driven from
    mfem_amd_pcg_pa_par with C callbacks, against the serial device solve."""
    _run_native(world, problem, p, n, mode, 1)


def test_cxx_transport_rccl_world_of_one():
    """The same library on RCCL (ncclCommInitRank, ncclAllReduce on the launch stream) with the only world one GPU
    allows; eight-GPU runs are the driver's."""
    _run_native(1, "bp3", 3, 5, "box", 0)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it starts two ranks itself (rehearsal on this one-GPU box: the
    ranks share the device and the exchange is staged through the host) and prints ONE JSON line with n_gpus = 2."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MFEM_AMD_DIRECT")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "1", "--elems", "12",
                        "--cg-iters", "10"], capture_output=True, text=True, timeout=280, env=env, cwd=str(ROOT))
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-2500:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 5 and out["config"]["numbering"] == "mfem"
    assert out["config"]["transport"] and out["cg"]["iters"] == 10
    assert out["config"]["element_interior_rows_written_by_element_kernel"] == 12**3*27      # as at N = 1
    assert out["config"]["global_dofs"] == (2*48 + 1)*49*49
