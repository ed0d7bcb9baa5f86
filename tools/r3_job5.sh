#!/bin/bash
# round 3, job 5: compute-bound fp64 MFMA probe; x staged between the halves of the pa_data requests (no spill) vs all at once
set -o pipefail
O=gpurun_out/r3job5; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for R in 1 16 64; do timeout -k 10 120 tools/mfma_probe 4096 100 $R >> $O/mfma_probe.txt 2>&1 || { cat $O/mfma_probe.txt; exit 1; }; done
cat $O/mfma_probe.txt
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mfma_stats -- tools/mfma_probe 4096 100 64 > $O/mfma_stats.log 2>&1 || { tail -5 $O/mfma_stats.log; exit 1; }
find $O/mfma_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/mfma_probe_kernel_stats.csv
cat $O/mfma_probe_kernel_stats.csv
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/mfma_pmc -- tools/mfma_probe 4096 20 64 > $O/mfma_pmc.log 2>&1 || { tail -5 $O/mfma_pmc.log; echo "pmc pass failed (continuing)"; }
python3 tools/pmc_summary.py $O/mfma_pmc > $O/mfma_pmc_summary.txt 2>&1; cat $O/mfma_pmc_summary.txt
timeout -k 10 600 python -u -m pytest tests/test_gpu_direct_rows.py tests/test_gpu_parity.py tests/test_gpu_parallel.py tests/test_gpu_cxx.py -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for lib in libmfem_amd_nosplit.so libmfem_amd.so libmfem_amd_nosplit.so libmfem_amd.so; do
  MFEM_AMD_LIB=$PWD/mfem_amd/$lib timeout -k 10 500 python -u bench.py --no-sizes --no-cpu --no-cxx > $O/bench_$lib.json 2>$O/bench_$lib.err || { tail -5 $O/bench_$lib.err; exit 1; }
  python3 -c "
import json,sys
r=json.loads(open('$O/bench_$lib.json').read().strip().splitlines()[-1])
print('$lib', 'apply ms', round(r['ms_per_step'],4), 'frac', round(r['roofline_apply']['frac'],4), 'l2e ms', round(r['roofline']['avg_launch_ms'],4), 'e2l ms', round(r['roofline_apply']['e2l_avg_launch_ms'],4), 'cg', round(r['cg']['hbm_frac_per_gpu'],4))
print(' by_order', [round(x['apply_frac'],3) for x in r['by_order']])
" | tee -a $O/summary.txt
done
