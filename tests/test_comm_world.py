"""CPU: the process-group bootstrap of libmfem_amd_comm.so (TCP rendezvous, pairwise sockets, host all-reduce,
barrier) with 3 ranks -- no GPU, no RCCL call (transport = TCP); and the library exports what
include/mfem_amd_comm.h declares."""
import os
import re
import sys
from pathlib import Path

import numpy as np
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def _worker(rank, world, port, out):
    try:
        from mfem_amd.comm import TRANSPORT_TCP, World
        w = World(rank, world, "127.0.0.1", port, 0, TRANSPORT_TCP)
        assert (w.rank, w.size, w.transport) == (rank, world, TRANSPORT_TCP)
        s = w.allreduce_host([rank + 1.0, 10.0*rank])
        m = w.allreduce_host([float(rank)], op="max")
        w.barrier()
        w.close()
        out.put((rank, s.tolist(), m.tolist()))
    except Exception:                                                        # noqa: BLE001
        import traceback
        out.put((rank, traceback.format_exc(), None))


def test_world_bootstrap_allreduce_barrier():
    world = 3
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 31000 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(30)
    for rank, s, m in res:
        assert m is not None, s
        assert s == [6.0, 30.0] and m == [2.0]


def test_comm_library_exports_every_declared_symbol():
    from mfem_amd import comm
    hdr = (ROOT / "include" / "mfem_amd_comm.h").read_text()
    declared = set(re.findall(r"\b(mfem_amd_(?:world|exchange)_\w+)\s*\(", hdr))
    assert declared and declared <= set(comm.EXPORTS), declared - set(comm.EXPORTS)
    for name in declared:
        getattr(comm.lib, name)
