"""Jacobi-PCG alone (BP3 p = 4 by default, ~1e7 dofs, interior-last numbering): for per-kernel traces."""
import sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from mfem_amd import capi
from mfem_amd.bp import BakeoffProblem
prob, p = (sys.argv[1], int(sys.argv[2])) if len(sys.argv) > 2 else ("bp3", 4)
dev = torch.device("cuda", 0)
capi.call("mfem_amd_init", 0)
n = max(1, round(1.0e7**(1/3)/p))
bp = BakeoffProblem(prob, p, (n, n, n), dev, device_tables=True, numbering="interior-last")
rhs = torch.rand(bp.ndofs, dtype=torch.float64, device=dev); rhs[bp.ess.long()] = 0.0
bp.pcg(rhs, 0.0, max_iter=100, check_every=100)
torch.cuda.synchronize()
t0 = time.perf_counter()
_, res, _ = bp.pcg(rhs, 0.0, max_iter=200, check_every=200)
torch.cuda.synchronize()
print("direct", bp.direct, "ms/iter", (time.perf_counter() - t0)/res.iterations*1e3)
