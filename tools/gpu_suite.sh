#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 1100 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/gpu_suite.log 2>&1
rc=$?
tail -4 gpurun_out/gpu_suite.log
exit $rc
