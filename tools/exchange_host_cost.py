"""Tuning aid: host time per call of the two transports of Exchange._transfer on RCCL with a world of one
(the only peer is the rank itself), to size the host-side cost of an exchange with 7 neighbours each way."""
import os, time
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29977")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
src = torch.arange(4096, dtype=torch.float64, device="cuda:0"); dst = torch.zeros_like(src)
def a2a():
    dist.all_to_all_single(dst, src, [4096], [4096])
def p2p(nops):
    ops = []
    for k in range(nops):
        ops.append(dist.P2POp(dist.irecv, dst[k*256:(k+1)*256], 0)); ops.append(dist.P2POp(dist.isend, src[k*256:(k+1)*256], 0))
    def f():
        for r in dist.batch_isend_irecv(ops):
            r.wait()
    return f
for name, fn in (("all_to_all_single", a2a), ("batch_isend_irecv x1 pair", p2p(1)), ("batch_isend_irecv x7 pairs", p2p(7))):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(300):
        fn()
    host = (time.perf_counter() - t0)/300
    torch.cuda.synchronize()
    tot = (time.perf_counter() - t0)/300
    print(f"{name}: host {host*1e6:.1f} us per call (with device drain {tot*1e6:.1f} us)", flush=True)
dist.destroy_process_group()
