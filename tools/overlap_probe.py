"""What would running the transpose-gather BESIDE the element kernel buy?  (timing only: the two streams race on the E-vector)
K element-kernel launches on one stream and K transpose-gather launches on another, together, against the same 2K launches
alternating on one stream (the apply as it is)."""
import sys, time, torch
sys.path.insert(0, ".")
from mfem_amd.bp import BakeoffProblem
import bench

def main():
    prob, p = sys.argv[1], int(sys.argv[2])
    dev = torch.device("cuda", 0)
    n = max(1, round((1.0e7)**(1.0/3.0)/p))
    bp = BakeoffProblem(prob, p, (n, n, n), dev, device_tables=True, numbering=bench.numbering_for((p*n + 1)**3, p))
    x = torch.rand(bp.ndofs, dtype=torch.float64, device=dev); y = torch.empty_like(x)
    sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    K = 60
    def seq():
        bp.use_stream(sA)
        for _ in range(K):
            bp.l2e(x, y=y); bp.e2l(y)
    def only(which):
        bp.use_stream(sA)
        for _ in range(K):
            (bp.l2e(x, y=y) if which == 0 else bp.e2l(y))
    def both():
        for _ in range(K):
            bp.use_stream(sA); bp.l2e(x, y=y)
            bp.use_stream(sB); bp.e2l(y)
    def timed(fn):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return (time.perf_counter() - t0)/K*1e6
    for fn in (seq, both):
        fn(); fn()
    r = {"seq": [], "both": [], "l2e": [], "e2l": []}
    for _ in range(3):
        r["seq"].append(timed(seq)); r["both"].append(timed(both))
        r["l2e"].append(timed(lambda: only(0))); r["e2l"].append(timed(lambda: only(1)))
    print(prob, "p", p, " ".join("%s %.1f" % (k, sorted(v)[1]) for k, v in r.items()), "us per apply", flush=True)

main()
