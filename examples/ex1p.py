#!/usr/bin/env python3
"""ex1p on MI355X: the parallel Poisson example of the reference (examples/ex1p.cpp, `-pa`) with one
process per GPU -- -Delta u = 1, homogeneous Dirichlet conditions, H1 order `-o`, partial assembly,
Jacobi-preconditioned CG (rel tol 1e-12, 2000 iterations at most, examples/ex1p.cpp:250-275).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
        examples/ex1p.py -m tests/golden/meshes/toroid-hex.mesh -o 3

Every rank reads the serial mesh's tables (as the reference builds ParMesh from the serial Mesh on
every rank), written once by the host-only tool examples/bin/dump_tables; the element partition is
the METIS-free chunk partition of mfem_amd/parallel.py (the reference calls METIS: the partition --
not the solution -- differs); shared dofs are exchanged over torch.distributed ("nccl" = RCCL over
xGMI; MFEM_AMD_DIST_BACKEND=gloo stages through the host, e.g. several ranks on one GPU).
Refinement follows ex1p.cpp:139-158: serial levels up to 10 000 elements, then two more ("parallel")
levels; `-r` gives the total number of levels instead."""
import argparse
import math
import os
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def mesh_ne(path):
    """number of elements in an MFEM mesh v1.0 file (the line after `elements`)"""
    with open(path) as f:
        lines = iter(f)
        for ln in lines:
            if ln.strip() == "elements":
                return int(next(lines).split()[0])
    raise SystemExit(f"{path}: no `elements` section")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-m", "--mesh", default=str(ROOT / "tests/golden/meshes/toroid-hex.mesh"))
    ap.add_argument("-o", "--order", type=int, default=1)
    ap.add_argument("-r", "--refine", type=int, default=-1, help="total uniform refinement levels (default: ex1p's rule)")
    ap.add_argument("-pa", "--partial-assembly", action="store_true", default=True)
    ap.add_argument("-d", "--device", default="hip")
    ap.add_argument("-no-vis", "--no-visualization", action="store_true")
    a = ap.parse_args()
    if a.device not in ("hip", "gpu"):
        raise SystemExit("ex1p.py: only -d hip (there is no CPU path)")

    import torch
    import torch.distributed as dist
    from mfem_amd import capi
    from mfem_amd.bp import BakeoffProblem
    from mfem_amd.parallel import Exchange, GeneralPartition, MeshTables, chunk_partition

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("MFEM_AMD_DIST_BACKEND", "nccl")
    if backend == "gloo":
        local_rank %= max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    capi.call("mfem_amd_init", local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group(backend, **({"device_id": dev} if backend == "nccl" else {}))

    levels = a.refine
    if levels < 0:
        ne = mesh_ne(a.mesh)
        levels = max(0, int(math.floor(math.log(10000.0/ne)/math.log(2.0)/3))) + 2      # ex1p.cpp:139-158
    # the serial tables, written once (rank 0) by the host-only C++ tool
    tmp = os.environ.get("MFEM_AMD_EX1P_DIR") or os.path.join(tempfile.gettempdir(), f"ex1p_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}")
    if rank == 0:
        os.makedirs(tmp, exist_ok=True)
        subprocess.check_call([str(ROOT / "examples/bin/dump_tables"), tmp, "file", a.mesh, str(levels), str(a.order), "2", "1"],
                              stdout=subprocess.DEVNULL)
    if world > 1:
        dist.barrier()
    rd = lambda n, dt: np.fromfile(os.path.join(tmp, n), dtype=dt)
    meta = rd("meta.i32", np.int32)
    NE, ndofs, D1D, Q1D, Dn, nn = int(meta[0]), int(meta[1]), int(meta[4]), int(meta[5]), int(meta[6]), int(meta[7])
    assert (D1D, Q1D) == (a.order + 1, a.order + 2)
    ngm = rd("nodes_gather_map.i32", np.int32).reshape(NE, Dn**3)
    nodes, byvdim = rd("nodes.f64", np.float64), int(rd("nodes_byvdim.i32", np.int32)[0])
    comp = lambda c: nodes[ngm*3 + c] if byvdim else nodes[c*nn + ngm]
    nodes_E = np.stack([comp(0), comp(1), comp(2)], axis=1)                       # (NE, 3, Dn^3) = E-vector (Dn^3,3,NE)
    tabs = MeshTables(D1D, ndofs, rd("gather_map.i32", np.int32), nodes_E, Dn, rd("ess_tdof.i32", np.int32))
    if world > 1:
        dist.barrier()
    if rank == 0 and not os.environ.get("MFEM_AMD_EX1P_DIR"):
        for f in os.listdir(tmp):
            os.unlink(os.path.join(tmp, f))
        os.rmdir(tmp)

    part = GeneralPartition(tabs, chunk_partition(NE, world), rank)
    ex = Exchange(part, dev) if world > 1 else None
    bp = BakeoffProblem("bp3", a.order, None, dev, part=part, exchange=ex)       # diffusion, Gauss-Legendre p+2 (ex1's rule)
    if rank == 0:
        print(f"Number of finite element unknowns: {ndofs}")
    b = bp.rhs_unit_source()
    X, res, hist = bp.pcg(b, 1e-12, max_iter=2000, check_every=8, want_hist=True)
    torch.cuda.synchronize()
    nrm2 = torch.zeros(1, dtype=torch.float64, device=dev)
    own = torch.from_numpy(~part.ghost).to(dev)
    nrm2[0] = torch.sum(X[own]**2)
    if world > 1:
        if backend == "gloo":
            h = nrm2.cpu(); dist.all_reduce(h); nrm2 = h
        else:
            dist.all_reduce(nrm2)
    if rank == 0:
        for i, v in enumerate(hist):
            print(f"   Iteration : {i:3d}  (B r, r) = {v:.6g}")
        if not res.converged:
            print("PCG: No convergence!")
        print(f"Average reduction factor = {(res.final_nom/res.initial_nom)**(0.5/max(1, res.iterations)):.6g}")
        print(f"ranks = {world}, elements = {NE}, iterations = {res.iterations}, |x|_2 = {float(nrm2.item())**0.5:.15e}")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
