#!/usr/bin/env python3
"""Stopping margins of the known-answer PCG cases: for every bake-off case of
tests/golden/baseline_anchors*.json, solved on the device at the reference probe's tolerance (rel 1e-12 on
sqrt((B r, r)), i.e. a 1e-24 reduction of (B r, r)), records the device iteration count beside the reference's,
(B r, r) one iteration before the earlier of the two stops, at it, and the threshold.  Writes
tests/golden/pcg_margins.json (run on the GPU box; the file is committed and read by the tests)."""
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))


def python_driver(cases, nfun):
    import torch
    from mfem_amd.bp import BakeoffProblem
    out = []
    for c in cases:
        p, n = c["p"], nfun(c)
        bp = BakeoffProblem(c["problem"], p, (n, n, n), "cuda:0")
        b = bp.rhs_unit_source()
        X, res, hist = bp.pcg(b, 1e-12, max_iter=2000, check_every=16, want_hist=True)
        torch.cuda.synchronize()
        k = min(res.iterations, c["pcg_its"])
        out.append({"driver": "python", "problem": c["problem"], "p": p, "its": int(res.iterations), "ref_its": c["pcg_its"],
                    "nom0": float(hist[0]), "nom_before": float(hist[k - 1]), "nom_at": float(hist[k]),
                    "threshold": float(hist[0])*1e-24})
        print(out[-1], flush=True)
        del bp, b, X
        torch.cuda.empty_cache()
    return out


def cxx_driver(bps, anchors, per_case_n=None):
    out = []
    for bp in bps:
        ref = {c["p"]: c for c in anchors["cases"] if c["problem"] == f"bp{bp}"}
        for p in range(1, 9):
            n = ref[p]["N"] if per_case_n else 0
            cmd = [str(ROOT / "examples" / "bin" / "bench_ceed"), "-bp", str(bp), "-p", str(p), "-n", str(n), "-reps", "1"]
            txt = subprocess.run(cmd, capture_output=True, text=True, timeout=900, check=True, cwd=str(ROOT)).stdout
            rows = [json.loads(l) for l in txt.splitlines() if l.startswith("{")]
            pr = [r for r in rows if r["phase"] == "problem"][0]
            out.append({"driver": "cxx", "problem": f"bp{bp}", "p": p, "its": pr["iterations"], "ref_its": ref[p]["pcg_its"],
                        "nom0": pr["nom0"], "nom_before": pr["nom_prev"], "nom_at": pr["nom_last"], "threshold": pr["threshold"]})
            print(out[-1], flush=True)
    return out


if __name__ == "__main__":
    import torch
    from mfem_amd import capi
    torch.cuda.init()
    capi.call("mfem_amd_init", 0)
    G = ROOT / "tests" / "golden"
    a = json.loads((G / "baseline_anchors.json").read_text())
    av = json.loads((G / "baseline_anchors_vec.json").read_text())
    rows = python_driver(a["cases"], lambda c: 48//c["p"]) + python_driver(av["cases"], lambda c: c["N"])
    rows += cxx_driver((1, 3, 5), a) + cxx_driver((2, 4, 6), av, per_case_n=True)
    doc = {"what": "device PCG iteration counts beside the reference's at rel tol 1e-12 (a 1e-24 reduction of (B r, r)), with "
                   "(B r, r) one iteration before the earlier stop (nom_before), at it (nom_at) and the threshold; cases with "
                   "its != ref_its are listed in `differ` and are the only ones the tests allow to differ",
           "generated_by": "tools/pcg_margins.py on MI355X",
           "cases": rows,
           "differ": [{"driver": r["driver"], "problem": r["problem"], "p": r["p"], "delta": r["its"] - r["ref_its"]}
                      for r in rows if r["its"] != r["ref_its"]]}
    (G / "pcg_margins.json").write_text(json.dumps(doc, indent=1) + "\n")
    (ROOT / "gpurun_out").mkdir(exist_ok=True)
    (ROOT / "gpurun_out" / "pcg_margins.json").write_text(json.dumps(doc, indent=1) + "\n")
    print("differ:", doc["differ"])
