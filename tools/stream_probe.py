"""Does the null stream cost anything?  The whole apply and the Jacobi-PCG, launched eagerly on the null stream
(torch's default) and on a created stream, and the apply replayed from a HIP graph.  One process, A/B/A order."""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from mfem_amd import capi
from mfem_amd.bp import BakeoffProblem

dev = torch.device("cuda", 0)
capi.call("mfem_amd_init", 0)
K = 40

def ev(fn, reps, stream):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1)*1e-3

for prob, p in (("bp3", 4), ("bp3", 6), ("bp3", 8), ("bp1", 4), ("bp3", 2)):
    n = max(1, round(1.0e7**(1/3)/p))
    bp = BakeoffProblem(prob, p, (n, n, n), dev, device_tables=True)
    x = torch.rand(bp.ndofs, dtype=torch.float64, device=dev); y = torch.empty_like(x)
    rhs = torch.rand(bp.ndofs, dtype=torch.float64, device=dev); rhs[bp.ess.long()] = 0.0
    null = torch.cuda.default_stream(dev)
    side = torch.cuda.Stream(dev)
    res = {}
    for tag, st in (("null", null), ("side", side), ("null2", null), ("side2", side)):
        bp.use_stream(st)
        with torch.cuda.stream(st):
            for _ in range(5):
                bp.mult(x, y)
            torch.cuda.synchronize()
            t = sorted(ev(lambda: bp.mult(x, y), K, st) for _ in range(5))[2]/K
            bp.pcg(rhs, 0.0, max_iter=3, check_every=3)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            bp.pcg(rhs, 0.0, max_iter=100, check_every=100)
            torch.cuda.synchronize()
            tc = (time.perf_counter() - t0)/100
        res[tag] = (t*1e6, tc*1e6)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        bp.use_stream(side)
        gr.capture_begin()
        for _ in range(K):
            bp.mult(x, y)
        gr.capture_end()
    gr.replay(); torch.cuda.synchronize()
    tg = sorted(ev(gr.replay, 1, torch.cuda.current_stream(dev)) for _ in range(5))[2]/K
    print(f"{prob} p={p} dofs={bp.ndofs}: " + "  ".join(f"{k}: apply {v[0]:.1f} us, pcg {v[1]:.1f} us/it" for k, v in res.items())
          + f"  graph: apply {tg*1e6:.1f} us", flush=True)
    del bp, x, y, rhs, gr
    torch.cuda.empty_cache()
