#!/usr/bin/env python3
"""bench.py -- BP3 (PA diffusion) operator-apply throughput on MI355X, plus Jacobi-PCG.

Contract (driver): `python bench.py --gpus N --steps K --warmup W` prints ONE JSON line.
A "step" is one L->L application of the PA diffusion operator (fused gather + sum-factorised
element kernel, then the deterministic transpose-gather), i.e. BilinearForm::Mult as
tests/benchmarks/bench_ceed.cpp `Kernel` times it; `value` = whole-job GDOF/s with all
inputs resident in HBM.  Workload at N=1: BASELINE config "CEED BP3 ... single MI355X",
p=4 (D1D=5, Q1D=6), 54^3 hexes = 1.02e7 dofs.  The same line carries the roofline of the
dominant kernel (measured live with HIP events), the whole-apply roofline, the PCG
DOFs*iter/s of the same operator and the CPU baseline (oracle port on the host cores).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec); ~6.3 TB/s achievable


def gl_rule(n):
    """Gauss-Legendre on [0,1] (numpy; host-side table for the benchmark inputs)."""
    x, w = np.polynomial.legendre.leggauss(n)
    return 0.5*(x + 1.0), 0.5*w


def gll_points(n):
    """Gauss-Lobatto points on [0,1]: endpoints + roots of P'_{n-1}."""
    if n == 2:
        return np.array([0.0, 1.0])
    c = np.zeros(n); c[-1] = 1.0
    r = np.polynomial.legendre.Legendre(c).deriv().roots()
    return 0.5*(np.concatenate(([-1.0], np.sort(r.real), [1.0])) + 1.0)


def lagrange_tables(nodes, pts):
    """B[q + Q*d], G[q + Q*d] of the Lagrange basis on `nodes` at `pts` (barycentric)."""
    n, Q = len(nodes), len(pts)
    B, G = np.zeros((n, Q)), np.zeros((n, Q))
    for d in range(n):
        others = np.delete(nodes, d)
        denom = np.prod(nodes[d] - others)
        for q, y in enumerate(pts):
            B[d, q] = np.prod(y - others)/denom
            s = 0.0
            for k in range(n - 1):
                s += np.prod(np.delete(y - others, k))
            G[d, q] = s/denom
    return B.reshape(-1).copy(), G.reshape(-1).copy()


def algorithmic_bytes(kind, D, Q, ne, ndofs):
    """bytes_apply = NE*(C*Q^3*8 + 2*D^3*4) + N_L*20; split per kernel."""
    ncomp = 6 if kind else 1
    elem = ne*(ncomp*Q**3*8 + D**3*4) + ndofs*8          # pa_data + gather_map + x   (L->E kernel)
    scat = ne*D**3*4 + ndofs*12                          # indices + offsets + y      (E->L kernel)
    return elem, scat


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--p", type=int, default=4)
    ap.add_argument("--n", type=int, default=0, help="elements per direction per GPU (default: ~1e7 dofs)")
    ap.add_argument("--problem", default="bp3", choices=["bp1", "bp3", "bp5"])
    ap.add_argument("--cg-iters", type=int, default=50)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from mfem_amd import capi
    from mfem_amd.cartesian import CartesianProblem

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    capi.call("mfem_amd_init", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    dev = torch.device("cuda", local_rank)

    p = args.p
    D = p + 1
    kind = 0 if args.problem == "bp1" else 1
    Q = p + 1 if args.problem == "bp5" else p + 2
    n = args.n or max(1, round((1.0e7)**(1.0/3.0)/p))
    # weak scaling: every rank owns an n^3 box of a (px,py,pz) process grid
    pg = {1: (1, 1, 1), 2: (2, 1, 1), 4: (2, 2, 1), 8: (2, 2, 2)}.get(world)
    if pg is None:
        raise SystemExit(f"unsupported world size {world}")
    rx, ry, rz = rank % pg[0], (rank//pg[0]) % pg[1], rank//(pg[0]*pg[1])
    prob = CartesianProblem(p, n, n, n, origin=(rx*n, ry*n, rz*n), global_n=(pg[0]*n, pg[1]*n, pg[2]*n))
    ne, ndofs = prob.ne, prob.ndofs

    nodes = gll_points(D)
    if args.problem == "bp5":
        qx = gll_points(Q)
        # GLL weights: 2/(n(n-1) P_{n-1}(x)^2) on [-1,1], halved for [0,1]
        c = np.zeros(Q); c[-1] = 1.0
        qw = 1.0/(Q*(Q - 1)*np.polynomial.legendre.Legendre(c)(2*qx - 1)**2)
    else:
        qx, qw = gl_rule(Q)
    B, G = lagrange_tables(nodes, qx)
    Bn, Gn = lagrange_tables(np.array([0.0, 1.0]), qx)
    W = (qw[None, None, :]*qw[None, :, None]*qw[:, None, None]).reshape(-1)

    t = lambda a, dt=None: torch.from_numpy(np.ascontiguousarray(a, dt)).to(dev)
    ptr = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None
    gm, off, ind = t(prob.gather_map), t(prob.offsets), t(prob.indices)
    nodesE, Wd = t(prob.nodes_E), t(W)
    LAYOUT = 1            # blocked pa_data layout (the product's own AssemblePA + AddMultPA)
    pa = torch.zeros(capi.lib.mfem_amd_pa_data_size(kind, LAYOUT, Q, ne), dtype=torch.float64, device=dev)
    one = torch.ones(1, dtype=torch.float64, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    capi.call("mfem_amd_pa_setup_from_nodes_3d", kind, 2, Q, 1, ne, Bn, Gn, ptr(Wd), ptr(nodesE), ptr(one), 1,
              LAYOUT, ptr(pa), None, stream)
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t0
    del nodesE

    g = torch.Generator(device="cpu"); g.manual_seed(1 + rank)
    x = torch.rand(ndofs, dtype=torch.float64, generator=g).to(dev)
    y = torch.empty_like(x)
    yE = torch.empty(ne*D**3, dtype=torch.float64, device=dev)

    def l2e():
        capi.call("mfem_amd_pa_apply_L2E", kind, LAYOUT, D, Q, ne, B, G, ptr(pa), ptr(gm), ptr(x), ptr(yE), stream)

    def e2l():
        capi.call("mfem_amd_restrict_scatter", ndofs, ptr(off), ptr(ind), ptr(yE), ptr(y), 0, 0, stream)

    def step():
        l2e(); e2l()

    def timed(fn, k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)*1e-3

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    total_dofs = ndofs*world
    value = total_dofs*args.steps/el/1e9

    # per-kernel durations (HIP events on the launch stream), same K
    t_l2e = timed(l2e, args.steps)/args.steps
    t_e2l = timed(e2l, args.steps)/args.steps
    b_l2e, b_e2l = algorithmic_bytes(kind, D, Q, ne, ndofs)
    roof = {"bound": "hbm", "kernel": "k_pa_apply (fused L->E gather + element kernel)",
            "achieved": b_l2e/t_l2e/1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": b_l2e/t_l2e/1e9/HBM_PEAK_GBS, "traffic": None,
            "algorithmic_bytes_per_launch": b_l2e, "avg_launch_ms": t_l2e*1e3}
    t_apply = el/args.steps
    roof_apply = {"bound": "hbm", "achieved": (b_l2e + b_e2l)/t_apply/1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": (b_l2e + b_e2l)/t_apply/1e9/HBM_PEAK_GBS, "algorithmic_bytes": b_l2e + b_e2l,
                  "e2l_avg_launch_ms": t_e2l*1e3, "e2l_achieved_GBs": b_e2l/t_e2l/1e9}

    # Jacobi-PCG on the constrained operator, exactly cg_iters iterations (rel_tol = 0)
    cg = None
    if world == 1 and args.cg_iters > 0:
        ess = prob.boundary_dofs()
        essd = t(ess)
        marker = torch.zeros(ndofs, dtype=torch.uint8, device=dev)
        masked = torch.empty(ne*D**3, dtype=torch.int32, device=dev)
        capi.call("mfem_amd_set_marker", len(ess), ptr(essd), ptr(marker), stream)
        capi.call("mfem_amd_mask_gather_map", ne*D**3, ptr(gm), ptr(marker), ptr(masked), stream)
        dE = torch.zeros(ne*D**3, dtype=torch.float64, device=dev)
        diag, dinv = torch.empty_like(x), torch.empty_like(x)
        if kind:
            capi.call("mfem_amd_pa_diffusion_diag_3d", D, Q, ne, 1, B, G, ptr(pa), LAYOUT, ptr(dE), stream)
        else:
            capi.call("mfem_amd_pa_mass_diag_3d", D, Q, ne, B, ptr(pa), LAYOUT, ptr(dE), stream)
        capi.call("mfem_amd_restrict_scatter", ndofs, ptr(off), ptr(ind), ptr(dE), ptr(diag), 0, 1, stream)
        capi.call("mfem_amd_jacobi_setup", ndofs, ptr(diag), 1.0, len(ess), ptr(essd), ptr(dinv), stream)
        del dE
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        op = capi.PaOperator(kind, LAYOUT, D, Q, ne, ndofs, vp(B), vp(G), ptr(pa), ptr(gm), ptr(masked), ptr(off), ptr(ind),
                             len(ess), ptr(essd), ptr(marker), ptr(yE))
        rhs = torch.rand(ndofs, dtype=torch.float64, generator=g).to(dev)
        rhs[essd.long()] = 0.0
        ws = torch.empty(capi.lib.mfem_amd_pcg_workspace_bytes(ndofs, args.cg_iters), dtype=torch.uint8, device=dev)
        sol = torch.empty_like(x)
        res = capi.PcgResult()

        def solve(iters):
            capi.call("mfem_amd_pcg_pa", C.byref(op), ptr(dinv), ptr(rhs), ptr(sol), 0.0, 0.0, iters, iters,
                      ptr(ws), C.byref(res), None, stream)

        solve(3)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        solve(args.cg_iters)
        torch.cuda.synchronize()
        tcg = time.perf_counter() - t0
        it = res.iterations
        cg_bytes = (b_l2e + b_e2l + 96*ndofs)*it
        cg = {"iters": it, "dofs_iter_per_s": ndofs*it/tcg, "ms_per_iter": tcg/it*1e3,
              "hbm_frac": cg_bytes/tcg/1e9/HBM_PEAK_GBS, "final_nom": res.final_nom, "initial_nom": res.initial_nom}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        cpu = cpu_baseline(kind, p, Q, B, G, Bn, Gn, W, args.cpu_seconds)

    if rank == 0:
        out = {"metric": "BP3 PA operator-apply GDOF/s (p=1..8) + CG DOFs*iter/s at 1/2/4/8 GPUs",
               "value": value, "unit": "GDOF/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": el/args.steps*1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"CEED {args.problem.upper()} PA operator apply (L->L), H1 p={p} "
                                      f"D1D={D} Q1D={Q}, Cartesian hex mesh {n}^3 per GPU",
                          "elements_per_gpu": ne, "dofs_per_gpu": ndofs, "process_grid": list(pg)},
               "roofline": roof, "roofline_apply": roof_apply, "cg": cg, "cpu_baseline": cpu,
               "setup_s": setup_s}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(kind, p, Q, B, G, Bn, Gn, W, seconds):
    """The CPU oracle (oracle/pa_oracle.c, the restatement of MFEM's `-d cpu` PA path) on the
    host cores: the same operator at the same order on a bounded sample (a 16^3 mesh),
    OpenMP over elements.  kind = "port"."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib as orc
    from mfem_amd.cartesian import CartesianProblem
    D = p + 1
    n = max(2, 64//p)
    prob = CartesianProblem(p, n, n, n)
    ne, ndofs = prob.ne, prob.ndofs
    J = np.zeros(9*Q**3*ne)
    orc.lib.orc_jacobians_3d(2, Q, ne, Bn, Gn, prob.nodes_E, J)
    c1 = np.ones(1)
    if kind:
        pa = np.zeros(6*Q**3*ne)
        orc.lib.orc_diffusion_setup_3d(Q, 1, ne, W, J, c1, 1, pa)
    else:
        detJ = np.zeros(Q**3*ne)
        orc.lib.orc_det_3d(Q, ne, J, detJ)
        pa = np.zeros(Q**3*ne)
        orc.lib.orc_mass_setup_3d(Q, ne, W, detJ, c1, 1, pa)
    del J
    op = orc.Operator(kind, D, Q, ne, ndofs, B, G, pa, prob.gather_map, prob.offsets, prob.indices)
    x = np.random.default_rng(1).random(ndofs)
    op.mult(x)
    reps, t0 = 0, time.perf_counter()
    while True:
        op.mult(x); reps += 1
        el = time.perf_counter() - t0
        if el > seconds:
            break
    return {"value": ndofs*reps/el/1e9, "unit": "GDOF/s", "cores": int(orc.lib.orc_num_threads()), "kind": "port",
            "sample": f"{reps} L->L applies of the same operator (p={p}, Q1D={Q}) on a {n}^3 mesh, {ndofs} dofs, "
                      f"{el:.1f} s, OpenMP over elements"}


if __name__ == "__main__":
    main()
