#!/bin/bash
# Two-rank rehearsal on one GPU (gloo + host-staged TCP transport from C++), rank 1 under rocprofv3 --kernel-trace: where the
# exchange kernels of an apply sit relative to the element kernel (both ranks are started by hand: nothing under the profiler
# launches another program)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29655 WORLD_SIZE=2 MFEM_AMD_DIST_BACKEND=gloo MFEM_AMD_BENCH_TRANSPORT=native-tcp
ARGS="--gpus 2 --steps 20 --warmup 5 --cg-iters 10"
# (rank 0 owns every shared dof of a two-rank split -- owner = lowest rank -- and has no ghost entries: the rank with the
# exchange to hide is rank 1, so that is the one traced)
RANK=0 LOCAL_RANK=0 timeout -k 10 300 python3 bench.py $ARGS > gpurun_out/ov_rank0.log 2>&1 &
p1=$!
RANK=1 LOCAL_RANK=1 timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ov_trace -- python3 bench.py $ARGS > gpurun_out/ov_rank1.log 2>&1
rc=$?
wait $p1 || rc=1
[ $rc -eq 0 ] || { tail -5 gpurun_out/ov_rank0.log gpurun_out/ov_rank1.log; exit 1; }
grep -a "^{" gpurun_out/ov_rank0.log | tail -1 | cut -c1-400
python3 tools/overlap_trace_report.py gpurun_out/ov_trace > gpurun_out/exchange_overlap_trace.txt
find gpurun_out/ov_trace -name "*.csv" -size +20M -delete
cat gpurun_out/exchange_overlap_trace.txt
