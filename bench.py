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
import math
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec); ~6.3 TB/s achievable
# HBM bytes per launch of the dominant kernel, from this round's PMC passes over THIS script (tools/pmc_bench.sh:
# FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc runs, per dispatch, averaged over the timed launches; gfx950
# correction of MI355X_MICROARCH.md, HBM section: FETCH_SIZE tallies the 128-B requests of wide streaming reads at
# 64 B -> doubled, WRITE_SIZE exact).  tools/pmc_summary.py writes bench_traffic.json next to this file with the
# workload key, the counters and the profile they come from; a workload without an entry reports null.
def measured_traffic(problem, p, n):
    f = ROOT / "bench_traffic.json"
    if not f.exists():
        return None, None
    ent = json.loads(f.read_text()).get(f"{problem}_p{p}_n{n}")
    return (ent["bytes_per_launch"], ent["source"]) if ent else (None, None)


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
    ap.add_argument("--elems", type=int, default=0, help="elements per direction per GPU (default: ~1e7 dofs per GPU)")
    ap.add_argument("--problem", default="bp3", choices=["bp1", "bp2", "bp3", "bp4", "bp5", "bp6"])
    ap.add_argument("--cg-iters", type=int, default=100)
    ap.add_argument("--no-orders", action="store_true", help="skip the p = 1..8 table (N=1 only; ~10 s)")
    ap.add_argument("--no-sizes", action="store_true", help="skip the BP1/BP3/BP5 x 1e5..1e8 dofs table (N=1 only; ~1 min)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-cxx", action="store_true", help="skip the C++ driver legs (examples/bin/bench_ceed, p = 4, 5, 6; N=1 only)")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N`: start the N ranks (one per GPU) before anything here touches the GPU
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist
    from mfem_amd import capi
    from mfem_amd.bp import BakeoffProblem
    from mfem_amd.parallel import BoxPartition, Exchange, process_grid

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MFEM_AMD_DIST_BACKEND") == "gloo":
        local_rank = local_rank % max(1, torch.cuda.device_count())   # rehearsal: ranks share GPUs
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback)")
    torch.cuda.set_device(local_rank)
    capi.call("mfem_amd_init", local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        backend = os.environ.get("MFEM_AMD_DIST_BACKEND", "nccl")   # "gloo": rehearsal of N ranks on one GPU
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    p = args.p
    n = args.elems or max(1, round((1.0e7)**(1.0/3.0)/p))
    # weak scaling: every rank owns an n^3 box of the (PX,PY,PZ) process grid; shared dofs on
    # the box interfaces are exchanged (P before, P^T after every operator application)
    pg = process_grid(world)
    t0 = time.perf_counter()
    if world > 1:
        # the same numbering class and therefore the same two kernels as at N = 1 (interior rows direct, packed E-vector)
        part = BoxPartition(p, (n, n, n), pg, rank, numbering=NUMBERING)
        vd = 3 if args.problem in ("bp2", "bp4", "bp6") else 1
        ex, transport = make_exchange(part, dev, vd if vd > 1 else 0, rank, world)
        bp = BakeoffProblem(args.problem, p, (n, n, n), dev, part=part, exchange=ex)
        owned = torch.tensor([part.n_owned*vd], dtype=torch.float64, device=dev)
        dist.all_reduce(owned)
        total_dofs = int(owned.item())
    else:
        transport = None
        bp = BakeoffProblem(args.problem, p, (n, n, n), dev, device_tables=True, numbering=numbering_for((p*n + 1)**3, p))
        total_dofs = bp.ndofs
    torch.cuda.synchronize()
    setup_s = time.perf_counter() - t0
    kind, D, Q, ne, ndofs = bp.kind, bp.D, bp.Q, bp.ne, bp.ndofs
    n_direct = bp.direct if bp.direct_active() else 0      # (bp is deleted before the JSON line is written)

    g = torch.Generator(device="cpu"); g.manual_seed(1 + rank)
    x = torch.rand(ndofs, dtype=torch.float64, generator=g).to(dev)
    if world > 1 and bp.ghost.numel():
        x[bp.ghost.long()] = 0.0
    y = torch.empty_like(x)

    def step():
        bp.mult(x, y)

    def timed(fn, k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(k):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)*1e-3

    b_l2e, b_e2l = algorithmic_bytes(kind, D, Q, ne, ndofs)
    nspin = spin_count(b_l2e + b_e2l)
    quiesce()
    spin_up(step, nspin)               # clock ramp after the idle set-up phase; then the W warm-up steps proper
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
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([el], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        el = float(tt.item())
    value = total_dofs*args.steps/el/1e9

    # the same K steps replayed from one HIP graph (N = 1; reported beside the eager figure, never as `value`: the
    # N > 1 lines are eager and the driver compares them with this one)
    replay = None
    if world == 1:
        side = torch.cuda.Stream(dev)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            bp.use_stream(side)
            gr.capture_begin()
            for _ in range(args.steps):
                step()
            gr.capture_end()
        bp.use_stream(torch.cuda.current_stream(dev))
        gr.replay(); torch.cuda.synchronize()
        quiesce()
        spin_up(gr.replay, max(1, nspin//args.steps))
        tr = sorted(timed(gr.replay, 1) for _ in range(3))[1]/args.steps
        replay = {"launch": f"hip graph of {args.steps} applies", "ms_per_step": tr*1e3, "GDOF/s": total_dofs/tr/1e9}
        del gr

    # per-kernel average launch durations on this rank (HIP events on the launch stream), K launches per
    # block; the median of three blocks (box-to-box and clock noise is +-5 % on a 17 ms block)
    quiesce()
    spin_up(step, nspin)
    t_l2e, t_e2l = kernel_times(bp, x, y, args.steps)
    roof = {"bound": "hbm", "kernel": "k_pa_apply (fused L->E gather + element kernel)",
            "achieved": b_l2e/t_l2e/1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": b_l2e/t_l2e/1e9/HBM_PEAK_GBS, "traffic": measured_traffic(args.problem, p, n)[0],
            "traffic_source": measured_traffic(args.problem, p, n)[1],
            "algorithmic_bytes_per_launch": b_l2e, "avg_launch_ms": t_l2e*1e3}
    t_apply = el/args.steps
    roof_apply = {"bound": "hbm", "achieved": (b_l2e + b_e2l)/t_apply/1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": (b_l2e + b_e2l)/t_apply/1e9/HBM_PEAK_GBS, "algorithmic_bytes_per_gpu": b_l2e + b_e2l,
                  "e2l_avg_launch_ms": t_e2l*1e3, "e2l_achieved_GBs": b_e2l/t_e2l/1e9}

    # Jacobi-PCG on the constrained operator, exactly cg_iters iterations (rel_tol = 0)
    cg = None
    if args.cg_iters > 0:
        rhs = torch.rand(ndofs, dtype=torch.float64, generator=g).to(dev)
        rhs[bp.ess.long()] = 0.0
        if world > 1 and bp.ghost.numel():
            rhs[bp.ghost.long()] = 0.0
        quiesce()
        bp.pcg(rhs, 0.0, max_iter=args.cg_iters, check_every=args.cg_iters)        # untimed: same solve, clocks up
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        _, res, _ = bp.pcg(rhs, 0.0, max_iter=args.cg_iters, check_every=cg_check(args.cg_iters))
        torch.cuda.synchronize()
        tcg = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([tcg], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            tcg = float(tt.item())
        it = res.iterations
        cg_bytes = (b_l2e + b_e2l + 96*ndofs)*it
        cg = {"iters": it, "dofs_iter_per_s": total_dofs*it/tcg, "ms_per_iter": tcg/it*1e3,
              "hbm_frac_per_gpu": cg_bytes/tcg/1e9/HBM_PEAK_GBS, "final_nom": res.final_nom, "initial_nom": res.initial_nom}

    # the metric's "p = 1..8": the same step at every order, ~1e7 dofs each, 20 steps (N = 1 only).  The
    # headline `value` stays the p = 4 workload above; this table rides along in the same JSON line.
    by_order = by_size = None
    if world == 1 and not args.no_orders:
        del bp, x, y
        torch.cuda.empty_cache()
        by_order = [size_row(args.problem, q, 1.0e7, dev, g, cg_iters=0) for q in range(1, 9)]
    # the metric's "1e5 to 1e8 DOFs" for BP1 / BP3 / BP5 (tests/benchmarks/bench_ceed.cpp:142-190) at the headline
    # order: whole apply and Jacobi-PCG per size.  Below 3e6 dofs the applies are replayed from one HIP graph
    # (kernel launches from Python cost more than the kernels there).
    if world == 1 and not args.no_sizes:
        by_size = []
        for prob in ("bp1", "bp3", "bp5"):
            for dofs in (1.0e5, 1.0e6, 1.0e7, 1.0e8):
                by_size.append(size_row(prob, p, dofs, dev, g, cg_iters=20))

    # the same problems through the C++ host layer and its drivers (the `ex1 / bake-off problems run unchanged through C++
    # host code` half of the north star): examples/bin/bench_ceed as child processes, after every timed region of this one
    cxx = None
    if rank == 0 and world == 1 and not args.no_cxx:
        torch.cuda.empty_cache()
        cxx = cxx_drivers(args.problem)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu:
        from mfem_amd.tables import bakeoff_tables
        cpu = cpu_baseline(args.problem, p, args.cpu_seconds)

    if rank == 0:
        out = {"metric": "BP3 PA operator-apply GDOF/s (p=1..8) + CG DOFs*iter/s at 1/2/4/8 GPUs",
               "value": value, "unit": "GDOF/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": el/args.steps*1e3, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": f"CEED {args.problem.upper()} PA operator apply (L->L, P^T A P across GPUs), H1 p={p} "
                                      f"D1D={D} Q1D={Q}, Cartesian hex mesh {n}^3 per GPU",
                          "elements_per_gpu": ne, "local_dofs_per_gpu": ndofs, "global_dofs": total_dofs,
                          "process_grid": list(pg), "transport": transport,
                          "numbering": numbering_for(ndofs, p),
                          "element_interior_rows_written_by_element_kernel": int(n_direct)},
               "roofline": roof, "roofline_apply": roof_apply, "graph_replay": replay, "cg": cg, "by_order": by_order, "by_size": by_size,
               "cxx_drivers": cxx, "cpu_baseline": cpu,
               "setup_s": setup_s}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


# L-dof numbering and element order of the boxes (every rank's box at N > 1):
#   "mfem" (default)  the reference's own: Mesh::MakeCartesian3D in its space-filling-curve element order, H1 dofs numbered
#                     vertex / edge / face / interior (fem/fespace.cpp) -- tables from the C++ host layer
#                     (mfem_amd/hosttables.py), i.e. what examples/bin/bench_ceed and ex1 run on;
#   "interior-last"   tensor-grid order of the vertex/edge/face dofs, element-interior dofs last, lexicographic elements;
#   "lexicographic"   the plain tensor-grid numbering (no reference caller produces it; kept for comparisons).
NUMBERING = os.environ.get("MFEM_AMD_BENCH_NUMBERING", "mfem")


def launch_ranks(n, argv):
    """One process per GPU through torch.distributed.run, as the driver's own command line does; the JSON line of rank 0 is
    relayed as the only stdout line.  With fewer than n devices visible (a rehearsal on a one-GPU box) the ranks share the
    devices and the exchange goes through the host (config.transport says so).  Returns the exit code."""
    import socket
    import subprocess
    import torch
    env = dict(os.environ)
    if torch.cuda.device_count() < n:                # (counting devices does not initialise the GPU)
        env.setdefault("MFEM_AMD_DIST_BACKEND", "gloo")
        env.setdefault("MFEM_AMD_BENCH_TRANSPORT", "native-tcp")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve())] + list(argv)
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    if r.returncode != 0 or not lines:
        sys.stderr.write(r.stdout)
        return r.returncode or 1
    print(lines[-1])
    return 0


def cxx_drivers(problem):
    """examples/bin/bench_ceed (the bake-off Kernel / Problem set-up of tests/benchmarks/bench_ceed.cpp on the C++ host layer)
    at p = 4, 5, 6, ~1e7 dofs: whole-apply and Jacobi-PCG fractions of the 8 TB/s roofline on SURVEY 8d's bytes."""
    import subprocess
    exe = ROOT / "examples" / "bin" / "bench_ceed"
    if not exe.exists():
        return {"error": f"{exe} is missing (make -C mfem_amd/host)"}
    rows = []
    for q in (4, 5, 6):
        cmd = [str(exe), "-bp", problem[2:], "-p", str(q), "-n", str(round(216/q)), "-reps", "50", "-mi", "50", "-rtol", "0"]
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        except subprocess.TimeoutExpired:
            rows.append({"p": q, "error": "timeout"})
            continue
        ph = {}
        for ln in r.stdout.splitlines():
            if ln.startswith("{"):
                d = json.loads(ln)
                ph[d["phase"]] = d
        if r.returncode != 0 or "kernel" not in ph or "problem" not in ph:
            rows.append({"p": q, "error": (r.stderr or r.stdout)[-300:]})
            continue
        k, pr = ph["kernel"], ph["problem"]
        rows.append({"p": q, "dofs": k["dofs"], "ms_per_apply": k["ms_per_apply"], "cxx_apply_frac": k["hbm_frac"],
                     "GDOF/s": k["MDof/s"]*1e-3, "pcg_iterations": pr["iterations"],
                     "ms_per_iter": pr["solve_s"]/max(1, pr["iterations"])*1e3, "cxx_pcg_frac": pr["hbm_frac"], "setup_s": k["setup_s"]})
    return {"driver": "examples/bin/bench_ceed -bp %s -p <p> -n <216/p> -reps 50 -mi 50 -rtol 0" % problem[2:], "rows": rows}


def kernel_times(bp, x, y, k):
    """Average duration of the two kernels of the apply where they run -- alternating, as inside mult(), with HIP events
    on the launch stream around each (median of three blocks of k applies).  Timed in a loop of its own, either kernel
    meets a cache state that no application ever has (profiles/r2/apply_trace_in_context.txt): the transpose-gather finds
    its E-vector cached, the element kernel re-writes lines that its previous launch left dirty."""
    import torch

    def block():
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(k)]
        for a, b, c in ev:
            a.record(); bp.l2e(x, y=y); b.record(); bp.e2l(y); c.record()
        torch.cuda.synchronize()
        return (sum(a.elapsed_time(b) for a, b, _ in ev)*1e-3/k, sum(b.elapsed_time(c) for _, b, c in ev)*1e-3/k)

    blocks = sorted(block() for _ in range(3))
    return blocks[1]


def numbering_for(ndofs, p):
    """one numbering for every row of every table (NUMBERING above); the library decides per case whether the interior
    rows go direct (vector_cg.hip: constrained_mult) -- the rows report it"""
    return NUMBERING


def spin_count(nbytes):
    """Untimed launches that precede every timed region: ~0.1 s of device work at 3 TB/s for a launch of `nbytes`
    algorithmic bytes.  After ANY idle period (set-up, a host-side pause, 0.5 s of sleep) the device needs 25-35 ms
    of continuous work to reach its steady rate -- the first 40-60 applies run 3-15 % slower
    (profiles/r2/warmup_probe.txt) -- and a 20-50 step timed block that starts cold is mostly that ramp.  The
    count depends on the workload only, so every rank of a multi-GPU run issues the same number of exchanges."""
    return int(min(4000, max(10, math.ceil(0.1/max(1.0e-9, nbytes/3.0e12)))))


def spin_up(fn, count):
    for _ in range(count):
        fn()


def cg_check(iters):
    """Iterations per host poll of the timed solves (MFEM_AMD_BENCH_CG_CHECK overrides)."""
    e = os.environ.get("MFEM_AMD_BENCH_CG_CHECK")
    return max(1, int(e)) if e else iters


def quiesce():
    """Before a timed region: run the cyclic garbage collector NOW and drain the device.  A collection that starts
    inside a timed loop frees the previous workload's tensors there, and releasing device memory stalls the queue for
    ~30 ms -- one such event inside a 20-launch loop reads as a 10x slower kernel (profiles/r2/stall_gc0.txt vs
    stall_gc1.txt, profiles/r2/slow_launch_trace_report.txt: every kernel of the slow loop has its normal duration)."""
    import gc
    import torch
    gc.collect()
    torch.cuda.synchronize()


def make_exchange(part, dev, vdim, rank, world):
    """The shared-dof exchange of a rank: the C++ RCCL transport (libmfem_amd_comm.so: ncclSend/ncclRecv/ncclAllReduce
    issued from C callbacks, nothing of a PCG iteration in the interpreter), checked here against the
    torch.distributed transport on one forward and one reverse exchange; all ranks fall back to the latter together if
    any of them could not set the former up (MFEM_AMD_BENCH_TRANSPORT=torch forces it)."""
    import torch
    import torch.distributed as dist
    from mfem_amd.parallel import Exchange
    ex_t = Exchange(part, dev, vdim=vdim)
    native, why = None, ""
    mode = os.environ.get("MFEM_AMD_BENCH_TRANSPORT", "native")      # "native-tcp": rehearsal of ranks sharing one GPU
    torch_name = "torch.distributed all_to_all_single (RCCL)"

    def all_ok(flag):
        ok = torch.tensor([1.0 if flag else 0.0], device=dev)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        return ok.item() > 0

    # stage 1: every rank builds its side (a rank that cannot must not leave the others waiting in a receive)
    if mode != "torch" and (dist.get_backend() == "nccl" or mode == "native-tcp"):
        try:
            from mfem_amd.comm import NativeExchange, World
            w = World(rank, world, os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500")), 23,
                      1 if mode == "native-tcp" else 0)
            native = NativeExchange(w, part, dev, vdim=vdim)
        except Exception as e:                                               # noqa: BLE001
            native, why = None, f"{type(e).__name__}: {e}"
    if not all_ok(native is not None):
        return ex_t, torch_name + (f"; native transport not used: {why}" if why else "")
    # stage 2: one forward and one reverse exchange on both transports, bit for bit
    nloc = part.prob.ndofs*max(1, vdim)
    g = torch.Generator(device="cpu"); g.manual_seed(77 + rank)
    v1 = torch.rand(nloc, dtype=torch.float64, generator=g).to(dev)
    ghost = torch.from_numpy(native.part.ext_ldof.astype("int64")).to(dev)
    v1[ghost] = 0.0
    v2 = v1.clone()
    same = False
    try:
        native.forward(v1); ex_t.forward(v2)
        same = bool(torch.equal(v1, v2))
        native.reverse(v1); ex_t.reverse(v2)
        torch.cuda.synchronize()
        same = same and bool(torch.equal(v1, v2))
    except Exception as e:                                                   # noqa: BLE001
        why = f"{type(e).__name__}: {e}"
    if all_ok(same):
        return native, ("host-staged TCP (rehearsal)" if mode == "native-tcp" else "RCCL") + " from C++ (libmfem_amd_comm.so), checked against torch.distributed"
    return ex_t, torch_name + "; native transport not used: " + (why or "it disagreed with torch.distributed on a test exchange")


def size_row(problem, p, dofs, dev, gen, cg_iters=20):
    """One (problem, order, size) line: whole L->L apply (and cg_iters Jacobi-PCG iterations) on an n^3 box with
    ~dofs unknowns; fractions of the 8 TB/s roofline on SURVEY 8d's algorithmic bytes."""
    import torch
    from mfem_amd.bp import BakeoffProblem
    n = max(1, round(dofs**(1.0/3.0)/p))
    bq = BakeoffProblem(problem, p, (n, n, n), dev, device_tables=True, numbering=numbering_for((p*n + 1)**3, p))
    xq = torch.rand(bq.ndofs, dtype=torch.float64, generator=gen).to(dev)
    yq = torch.empty_like(xq)
    b1, b2 = algorithmic_bytes(bq.kind, bq.D, bq.Q, bq.ne, bq.ndofs)
    graph = bq.ndofs < 3_000_000 or os.environ.get("MFEM_AMD_BENCH_GRAPH") == "1"      # (forced: the 1e7-dof comparison of DESIGN 5)
    # enough applies per timed block for >= ~4 ms of device work
    k = int(min(400, max(20, 4.0e-3/max(2.0e-6, (b1 + b2)/5.0e12))))

    def ev_time(fn, reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)*1e-3

    for _ in range(3):
        bq.mult(xq, yq)
    quiesce()
    nspin = spin_count(b1 + b2)
    spin_up(lambda: bq.mult(xq, yq), nspin)
    if graph:
        side = torch.cuda.Stream(dev)
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            bq.use_stream(side)
            gr.capture_begin()
            for _ in range(k):
                bq.mult(xq, yq)
            gr.capture_end()
        bq.use_stream(torch.cuda.current_stream(dev))
        spin_up(gr.replay, max(1, nspin//k))
        tq = sorted(ev_time(gr.replay, 1) for _ in range(5))[2]/k
    else:
        tq = sorted(ev_time(lambda: bq.mult(xq, yq), k) for _ in range(3))[1]/k
    tl = kernel_times(bq, xq, yq, 20)[0]
    row = {"problem": problem, "p": p, "n": n, "dofs": bq.ndofs, "GDOF/s": bq.ndofs/tq/1e9, "ms_per_step": tq*1e3,
           "apply_frac": (b1 + b2)/tq/1e9/HBM_PEAK_GBS, "l2e_ms": tl*1e3, "l2e_frac": b1/tl/1e9/HBM_PEAK_GBS,
           "launch": f"hip graph of {k} applies" if graph else "eager", "numbering": numbering_for(bq.ndofs, p),
           "interior_rows_direct": bq.direct if bq.direct_active() else 0}
    if cg_iters > 0:
        rhs = torch.rand(bq.ndofs, dtype=torch.float64, generator=gen).to(dev)
        rhs[bq.ess.long()] = 0.0
        quiesce()
        spin_up(lambda: bq.mult(xq, yq), nspin)
        bq.pcg(rhs, 0.0, max_iter=cg_iters, check_every=cg_iters)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        _, res, _ = bq.pcg(rhs, 0.0, max_iter=cg_iters, check_every=cg_check(cg_iters))
        torch.cuda.synchronize()
        tcg = time.perf_counter() - t0
        row["cg_dofs_iter_per_s"] = bq.ndofs*res.iterations/tcg
        row["cg_frac"] = (b1 + b2 + 96*bq.ndofs)*res.iterations/tcg/1e9/HBM_PEAK_GBS
    del bq, xq, yq
    torch.cuda.empty_cache()
    return row


def cpu_baseline(problem, p, seconds):
    """The CPU oracle (oracle/pa_oracle.c, the restatement of MFEM's `-d cpu` PA path) on the
    host cores: the same operator at the same order on a bounded sample -- a mesh of ~1e6 dofs, whose pa_data
    (160 MB at p = 4) does not fit the host's last-level cache -- OpenMP over elements.  kind = "port"."""
    sys.path.insert(0, str(ROOT / "tests"))
    import oracle_lib as orc
    from mfem_amd.cartesian import CartesianProblem
    from mfem_amd.tables import bakeoff_tables
    kind, D, Q, B, G, Bn, Gn, W = bakeoff_tables(problem, p)
    n = max(2, round(1.0e6**(1.0/3.0)/p))
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
    try:
        import psutil
        phys, logical = psutil.cpu_count(logical=False), psutil.cpu_count(logical=True)
    except Exception:                                                        # noqa: BLE001
        phys, logical = None, os.cpu_count()
    out = {"value": ndofs*reps/el/1e9, "unit": "GDOF/s", "cores": int(orc.lib.orc_num_threads()), "kind": "port",
           "host_physical_cores": phys, "host_logical_cpus": logical,
           "sample": f"{reps} L->L applies of the same operator (p={p}, Q1D={Q}) on a {n}^3 mesh, {ndofs} dofs, "
                     f"{el:.1f} s, OpenMP over elements"}
    # one core on the mesh of BASELINE.md's probe (48/p elements per direction, ~1.2e5 dofs, cache-resident): the figure that
    # compares with the compiled reference's single-core numbers there (13.1 MDOF/s at BP3 p = 4 in the survey container)
    nthreads = int(orc.lib.orc_num_threads())
    try:
        orc.lib.orc_set_num_threads(1)
        ns = max(2, 48//p)
        ps = CartesianProblem(p, ns, ns, ns)
        Js = np.zeros(9*Q**3*ps.ne)
        orc.lib.orc_jacobians_3d(2, Q, ps.ne, Bn, Gn, ps.nodes_E, Js)
        if kind:
            pas = np.zeros(6*Q**3*ps.ne); orc.lib.orc_diffusion_setup_3d(Q, 1, ps.ne, W, Js, c1, 1, pas)
        else:
            ds = np.zeros(Q**3*ps.ne); orc.lib.orc_det_3d(Q, ps.ne, Js, ds)
            pas = np.zeros(Q**3*ps.ne); orc.lib.orc_mass_setup_3d(Q, ps.ne, W, ds, c1, 1, pas)
        ops = orc.Operator(kind, D, Q, ps.ne, ps.ndofs, B, G, pas, ps.gather_map, ps.offsets, ps.indices)
        xs = np.random.default_rng(2).random(ps.ndofs)
        ops.mult(xs)
        r1, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < 2.0:
            ops.mult(xs); r1 += 1
        e1 = time.perf_counter() - t0
        out["one_core"] = {"MDOF/s": ps.ndofs*r1/e1/1e6, "sample": f"{r1} applies on a {ns}^3 mesh ({ps.ndofs} dofs), one thread, {e1:.1f} s",
                           "reference_in_survey_container_MDOF/s": {"bp3": [3.2, 7.3, 8.2, 13.1, 14.7, 13.6, 14.7, 11.8],
                                                                    "bp1": [9.9, 21.5, 31.4, 35.3, 37.2, 38.5, 38.4, 26.1],
                                                                    "bp5": [7.4, 14.0, 17.0, 17.9, 20.3, 18.7, 17.0, 14.2]}.get(problem, [None]*8)[p - 1]}
    finally:
        orc.lib.orc_set_num_threads(nthreads)
    if kind:
        # the same Jacobi-PCG as the `cg` object (unit source, all-boundary Dirichlet), a fixed number of iterations.  The
        # apply is threaded over elements; the vector updates and dot products are threaded over contiguous blocks, one
        # per core, partial sums added in block order (what MFEM's -d cpu path does with one MPI rank per core; the sum
        # order differs from the sequential single-rank reference: a timing leg, not a parity one)
        orc.lib.orc_set_threaded_vectors(1)
        ess = prob.boundary_dofs().astype(np.int32)
        opc = orc.Operator(kind, D, Q, ne, ndofs, B, G, pa, prob.gather_map, prob.offsets, prob.indices, ess=ess)
        dinv = orc.jacobi_setup(opc.diag(), ess)
        b = np.ones(ndofs); b[ess] = 0.0
        # at least ~1.5 s of Jacobi-PCG: a short first leg sizes the sample
        t0 = time.perf_counter()
        opc.pcg(b, dinv, 0.0, max_iter=5)
        per_it = (time.perf_counter() - t0)/5
        its_cap = int(min(400, max(20, 1.5/max(per_it, 1e-4))))
        t0 = time.perf_counter()
        _, its, _, _ = opc.pcg(b, dinv, 0.0, max_iter=its_cap)
        elc = time.perf_counter() - t0
        orc.lib.orc_set_threaded_vectors(0)
        out["cg_dofs_iter_per_s"] = ndofs*its/elc
        out["cg_sample"] = f"{its} Jacobi-PCG iterations on the same mesh, {elc:.2f} s, vector updates and dots threaded per core"
    return out


if __name__ == "__main__":
    main()
