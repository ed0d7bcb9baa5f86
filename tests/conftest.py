import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def golden_cases(pattern="*"):
    return sorted(p.stem for p in GOLDEN.glob(pattern + ".npz"))


def load_case(name):
    z = np.load(GOLDEN / (name + ".npz"))
    d = {k: z[k] for k in z.files}
    for k in ("p", "kind", "rule", "varcoef", "D1D", "Q1D", "NE", "ndofs", "nodes_D1D", "nodes_byvdim",
              "nodes_ndofs", "cg_iters", "cg_converged", "nedges", "nfaces"):
        d[k] = int(d[k][0])
    return d


def splitmix(seed, n):
    """Build-owned input stream (same as oracle/ref_dump.cpp smix): n values in [0,1)."""
    M = (1 << 64) - 1
    out = np.empty(n)
    i = np.arange(1, n + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64((seed*0x9E3779B97F4A7C15) & M) + i*np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(30)))*np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27)))*np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    out[:] = (z >> np.uint64(11)).astype(np.float64)*(1.0/9007199254740992.0)
    return out
