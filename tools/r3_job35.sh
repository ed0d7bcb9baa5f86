#!/bin/bash
# full-size rehearsal of the driver's N > 1 command on ONE GPU: 4 ranks share the device (gloo + host-staged TCP), default
# 1e7 dofs per rank; checks set-up time / memory of the partitioned path at the size the 8-GPU run uses
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python bench.py --gpus 4 --steps 5 --warmup 2 --cg-iters 20 > gpurun_out/bench_g4_fullsize.json 2> gpurun_out/bench_g4_fullsize.err
rc=$?
tail -c 1500 gpurun_out/bench_g4_fullsize.json; tail -5 gpurun_out/bench_g4_fullsize.err
exit $rc
