#!/bin/bash
# rehearsal of bench.py --gpus 2 on ONE GPU: gloo process group, native C++ exchange over the host-staged TCP transport
export MFEM_AMD_DIST_BACKEND=gloo MFEM_AMD_BENCH_TRANSPORT=native-tcp
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 5 --warmup 2 --elems 24 --cg-iters 20 > gpurun_out/bench_g2_native_tcp.json 2> gpurun_out/bench_g2_native_tcp.err
echo "rc=$?"
export MFEM_AMD_BENCH_TRANSPORT=torch
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --steps 5 --warmup 2 --elems 24 --cg-iters 20 > gpurun_out/bench_g2_torch.json 2> gpurun_out/bench_g2_torch.err
echo "rc=$?"
