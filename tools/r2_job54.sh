#!/bin/bash
set -o pipefail
O=gpurun_out/job54; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests/test_gpu_direct_rows.py tests/test_gpu_parity.py tests/test_gpu_cxx.py tests/test_gpu_bakeoff.py -x -q -m gpu --timeout 300 --timeout-method thread > $O/tests.log 2>&1
rc=$?
tail -25 $O/tests.log
exit $rc
