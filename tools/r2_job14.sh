#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_bakeoff.py tests/test_gpu_vector.py tests/test_gpu_vector_ref.py tests/test_gpu_matcoef.py tests/test_gpu_parallel.py tests/test_gpu_cxx.py -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_pcg.log 2>&1
rc=$?
tail -5 gpurun_out/gpu_pcg.log
[ $rc -eq 0 ] || exit $rc
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/stats14; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --no-cpu --no-orders --no-sizes --steps 20 --warmup 5 > $out/bench.log 2>&1 || { tail -5 $out/bench.log; exit 1; }
tail -1 $out/bench.log
f=$(find $out -name "*kernel_stats.csv" | head -1); cut -c1-60,150- $f | head -8
