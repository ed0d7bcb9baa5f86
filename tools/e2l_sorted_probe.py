"""Tuning aid: time an E->L pass over an E-vector stored in L-dof order (contiguous segmented sum) and the
scattered store that layout would move into the element kernel, next to the shipped row-driven kernel.
    python tools/e2l_sorted_probe.py [p ...]"""
import ctypes as C
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from mfem_amd import capi                                    # noqa: E402
from mfem_amd.cartesian import CartesianProblem              # noqa: E402

lib = C.CDLL(str(ROOT / "tools/libe2l_sorted.so"))
vp, i64 = C.c_void_p, C.c_int64
lib.probe_segsum.argtypes = [i64, vp, vp, vp]
lib.probe_permute.argtypes = [i64, vp, vp, vp]
lib.probe_copy.argtypes = [i64, vp, vp]
lib.probe_rows.argtypes = [C.c_int, i64, vp, vp, vp, vp]
capi.call("mfem_amd_init", 0)
dev = "cuda:0"
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
P = lambda x: vp(x.data_ptr())


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/reps*1e3


for p in [int(a) for a in sys.argv[1:]] or [4]:
    n = round((1.0e7**(1/3) - 1)/p)
    prob = CartesianProblem(p, n, n, n)
    D3, ne, nd = (p + 1)**3, prob.ne, prob.ndofs
    off, ind = t(prob.offsets), t(prob.indices)
    pos = np.empty(ne*D3, np.int32); pos[prob.indices] = np.arange(ne*D3, dtype=np.int32)
    posd = t(pos)
    yE = torch.rand(ne*D3, dtype=torch.float64, device=dev)
    yS, y0, y1 = torch.empty_like(yE), torch.empty(nd, dtype=torch.float64, device=dev), torch.empty(nd, dtype=torch.float64, device=dev)
    assert lib.probe_permute(ne*D3, P(posd), P(yE), P(yS)) == 0
    capi.call("mfem_amd_restrict_scatter", nd, P(off), P(ind), P(yE), P(y0), 0, 0, None)
    assert lib.probe_segsum(nd, P(off), P(yS), P(y1)) == 0
    torch.cuda.synchronize()
    same = bool(torch.equal(y0, y1))
    t_row = timed(lambda: capi.call("mfem_amd_restrict_scatter", nd, P(off), P(ind), P(yE), P(y0), 0, 0, None))
    t_seg = timed(lambda: lib.probe_segsum(nd, P(off), P(yS), P(y1)))
    t_perm = timed(lambda: lib.probe_permute(ne*D3, P(posd), P(yE), P(yS)))
    t_copy = timed(lambda: lib.probe_copy(ne*D3, P(yE), P(yS)))
    extra = ""
    for nbp in [int(v) for v in __import__('os').environ.get('E2L_VARIANTS', '4,8,0').split(',')]:
        y2 = torch.empty_like(y0)
        assert lib.probe_rows(nbp, nd, P(off), P(ind), P(yE), P(y2)) == 0
        torch.cuda.synchronize()
        ok = bool(torch.equal(y0, y2))
        extra += f" | rows x{nbp} in flight {timed(lambda: lib.probe_rows(nbp, nd, P(off), P(ind), P(yE), P(y2))):.1f} us ({ok})"
    print(f"p={p} n={n} dofs={nd}: row-driven E->L {t_row:.1f} us | segmented sum {t_seg:.1f} us (bit-identical {same}) | "
          f"scattered store {t_perm:.1f} us vs coalesced copy {t_copy:.1f} us ({ne*D3*8/1e6:.0f} MB E-vector)" + extra, flush=True)
