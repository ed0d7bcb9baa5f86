"""How long after an idle period does the apply run below its steady rate?  Consecutive blocks of 10 applies, timed
with events, after (a) problem set-up + 3 applies + gc/sync (what bench.py does), (b) a 0.5 s sleep."""
import sys, time, gc, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from mfem_amd import capi
from mfem_amd.bp import BakeoffProblem

dev = torch.device("cuda", 0)
capi.call("mfem_amd_init", 0)

def blocks(bp, x, y, nblk, per):
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(nblk + 1)]
    evs[0].record()
    for b in range(nblk):
        for _ in range(per):
            bp.mult(x, y)
        evs[b + 1].record()
    torch.cuda.synchronize()
    return [evs[b].elapsed_time(evs[b + 1])*1e3/per for b in range(nblk)]

for prob, p in (("bp3", 4), ("bp3", 6), ("bp1", 4)):
    n = max(1, round(1.0e7**(1/3)/p))
    bp = BakeoffProblem(prob, p, (n, n, n), dev, device_tables=True)
    x = torch.rand(bp.ndofs, dtype=torch.float64, device=dev); y = torch.empty_like(x)
    for _ in range(3):
        bp.mult(x, y)
    gc.collect(); torch.cuda.synchronize()
    a = blocks(bp, x, y, 16, 10)
    time.sleep(0.5)
    b = blocks(bp, x, y, 16, 10)
    time.sleep(3.0)
    c = blocks(bp, x, y, 16, 10)
    print(f"{prob} p={p} us/apply per block of 10")
    print("  after set-up:   " + " ".join(f"{t:.0f}" for t in a))
    print("  after 0.5 s idle:" + " ".join(f"{t:.0f}" for t in b))
    print("  after 3 s idle:  " + " ".join(f"{t:.0f}" for t in c), flush=True)
    del bp, x, y
    torch.cuda.empty_cache()
