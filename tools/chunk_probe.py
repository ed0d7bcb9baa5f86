#!/usr/bin/env python3
"""Probe (tuning aid): chunk-pipelined L->L apply.  The element kernel runs over contiguous element
chunks that write ONE slot of a small E-vector ring; the sub-row E->L kernel sums the chunk's rows while
the slot is still cache-resident (Infinity Cache 256 MiB; MI355X_MICROARCH.md, HBM section).  Prints the
time of the whole apply against the two-kernel apply for several chunk sizes, ring sizes and one or two
streams, and checks that y is bit-identical."""
import argparse
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench  # noqa: E402


def chunk_tables(offsets, indices, D3, C_el):
    """sub-row tables sorted by (chunk, row): rows(codes), sub_off, rel indices, per-chunk sub-row ranges"""
    nrows = len(offsets) - 1
    counts = np.diff(offsets).astype(np.int64)
    row_of = np.repeat(np.arange(nrows, dtype=np.int64), counts)
    ind = indices.astype(np.int64)
    chunk_of = ind//(D3*C_el)
    key = chunk_of*nrows + row_of
    order = np.argsort(key, kind="stable")
    key_s = key[order]
    first = np.ones(len(key_s), bool)
    first[1:] = key_s[1:] != key_s[:-1]
    starts = np.flatnonzero(first)
    sub_chunk = key_s[starts]//nrows
    sub_row = key_s[starts] % nrows
    sub_off = np.append(starts, len(key_s)).astype(np.int32)
    rel = (ind[order] - chunk_of[order]*(C_el*D3)).astype(np.int32)
    assert counts.min() > 0, "rows without entries are not handled by this probe"
    first_chunk = chunk_of[offsets[:-1]]
    last_chunk = chunk_of[offsets[1:] - 1]
    code = sub_row.copy()
    code |= np.where(sub_chunk != first_chunk[sub_row], 1 << 30, 0)
    code |= np.where(sub_chunk != last_chunk[sub_row], 1 << 31, 0)
    code = (code & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
    nchunks = int(chunk_of.max()) + 1
    bounds = np.searchsorted(sub_chunk, np.arange(nchunks + 1))
    return code, sub_off, rel, bounds


def run(problem, p, n, steps, chunk_list):
    import torch
    from mfem_amd import capi
    from mfem_amd.bp import BakeoffProblem
    dev = torch.device("cuda", 0)
    bp = BakeoffProblem(problem, p, (n, n, n), dev)
    kind, D, Q, ne, ndofs = bp.kind, bp.D, bp.Q, bp.ne, bp.ndofs
    D3 = D**3
    nbz = capi.lib.mfem_amd_pa_data_size(kind, 1, Q, 1)//((6 if kind else 1)*Q**3)
    x = torch.rand(ndofs, dtype=torch.float64, device=dev)
    y0 = torch.empty_like(x)
    y1 = torch.empty_like(x)
    ptr = lambda t, off=0: C.c_void_p(t.data_ptr() + off)
    b1, b2 = bench.algorithmic_bytes(kind, D, Q, ne, ndofs)

    def timed(fn, k):
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(k):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1)*1e-3/k

    t_ref = timed(lambda: bp.mult(x, y0), steps)
    t_l2e = timed(lambda: bp.l2e(x), steps)
    t_e2l = timed(lambda: bp.e2l(y0), steps)
    bp.mult(x, y0)
    print(f"{problem} p={p} n={n} ne={ne} ndofs={ndofs}: two-kernel apply {t_ref*1e6:7.1f} us "
          f"(l2e {t_l2e*1e6:6.1f} + e2l {t_e2l*1e6:6.1f}) = {(b1+b2)/t_ref/8e12*100:4.1f}% of 8 TB/s", flush=True)

    off_h, ind_h = bp.prob.offsets, bp.prob.indices
    for C_el in chunk_list:
        C_el = max(nbz, (C_el//nbz)*nbz)
        code, sub_off, rel, bounds = chunk_tables(off_h, ind_h, D3, C_el)
        nchunks = len(bounds) - 1
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        code_d, soff_d, rel_d = t(code), t(sub_off), t(rel)
        for nslots, two_streams in ((nchunks, False), (1, False), (2, False), (2, True), (3, True)):
            ring = torch.empty(min(nslots, nchunks)*C_el*D3, dtype=torch.float64, device=dev)
            sA = torch.cuda.current_stream(dev)
            sS = torch.cuda.Stream(dev) if two_streams else sA
            evA = [torch.cuda.Event() for _ in range(nchunks)]
            evS = [torch.cuda.Event() for _ in range(nchunks)]

            def apply_chunked():
                pA, pS = C.c_void_p(sA.cuda_stream), C.c_void_p(sS.cuda_stream)
                if two_streams:
                    # fork: the previous apply ended with sA waiting for sS, so sA orders everything
                    evF = torch.cuda.Event(); evF.record(sA)
                    sS.wait_event(evF)
                for c in range(nchunks):
                    e0, e1 = c*C_el, min(ne, (c + 1)*C_el)
                    slot = c % nslots
                    if two_streams and c >= nslots:
                        sA.wait_event(evS[c - nslots])
                    pa_off = capi.lib.mfem_amd_pa_data_size(kind, 1, Q, e0)
                    capi.call("mfem_amd_pa_apply_L2E", kind, 1, D, Q, e1 - e0, bp.B, bp.G, ptr(bp.pa, 8*pa_off),
                              ptr(bp.gm, 4*e0*D3), ptr(x), ptr(ring, 8*slot*C_el*D3), pA)
                    if two_streams:
                        evA[c].record(sA)
                        sS.wait_event(evA[c])
                    s0, s1 = int(bounds[c]), int(bounds[c + 1])
                    capi.call("mfem_amd_restrict_scatter_subrows", s1 - s0, ptr(code_d, 4*s0), ptr(soff_d, 4*s0),
                              ptr(rel_d), ptr(ring, 8*slot*C_el*D3), ptr(y1), pS)
                    if two_streams:
                        evS[c].record(sS)
                if two_streams:
                    sA.wait_event(evS[nchunks - 1])

            y1.fill_(-7.0)
            apply_chunked()
            torch.cuda.synchronize()
            same = bool(torch.equal(y0, y1))
            tc = timed(apply_chunked, steps)
            tag = f"C={C_el:6d} ({nchunks:3d} chunks) slots={min(nslots, nchunks):3d} streams={2 if two_streams else 1}"
            print(f"   {tag}: {tc*1e6:7.1f} us = {(b1+b2)/tc/8e12*100:4.1f}%  subrows/rows={len(code)/ndofs:.3f} "
                  f"bit-identical={same}", flush=True)
            # graph replay of the same launch sequence
            if two_streams and nslots == 2:
                try:
                    g = torch.cuda.CUDAGraph()
                    cs = torch.cuda.Stream(dev)
                    with torch.cuda.stream(cs):
                        sA_old = sA
                        sA = cs
                        g.capture_begin()
                        apply_chunked()
                        g.capture_end()
                        sA = sA_old
                    torch.cuda.synchronize()
                    y1.fill_(-7.0)
                    g.replay()
                    torch.cuda.synchronize()
                    same = bool(torch.equal(y0, y1))
                    tg = timed(g.replay, steps)
                    print(f"   {tag} GRAPH: {tg*1e6:7.1f} us = {(b1+b2)/tg/8e12*100:4.1f}% bit-identical={same}", flush=True)
                except Exception as ex:      # noqa: BLE001
                    print(f"   graph capture failed: {ex!r}", flush=True)
                    torch.cuda.synchronize()
            del ring
        del code_d, soff_d, rel_d


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--problem", default="bp3")
    ap.add_argument("--p", default="4")
    ap.add_argument("--dofs", type=float, default=1e7)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--chunks", default="2940,5880,8820,11760,17640,26460")
    a = ap.parse_args()
    import torch
    from mfem_amd import capi
    torch.cuda.init()
    capi.call("mfem_amd_init", 0)
    for prob in a.problem.split(","):
        for p in [int(s) for s in a.p.split(",")]:
            n = max(1, round(a.dofs**(1/3)/p))
            run(prob, p, n, a.steps, [int(c) for c in a.chunks.split(",")])
