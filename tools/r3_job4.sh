#!/bin/bash
# round 3, job 4: fp64 MFMA vs vector FMA probe (rocprofv3 stats + counters); XCD-aware block orders with the packed E-vector
set -o pipefail
O=gpurun_out/r3job4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 120 tools/mfma_probe 4096 200 > $O/mfma_probe.txt 2>&1 || { cat $O/mfma_probe.txt; exit 1; }
cat $O/mfma_probe.txt
timeout -k 10 120 tools/mfma_probe 65536 50 >> $O/mfma_probe.txt 2>&1 || { cat $O/mfma_probe.txt; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mfma_stats -- tools/mfma_probe 4096 200 > $O/mfma_stats.log 2>&1 || { tail -5 $O/mfma_stats.log; exit 1; }
find $O/mfma_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/mfma_probe_kernel_stats.csv
cat $O/mfma_probe_kernel_stats.csv
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d $O/mfma_pmc -- tools/mfma_probe 4096 20 > $O/mfma_pmc.log 2>&1 || { tail -5 $O/mfma_pmc.log; echo "pmc pass failed (continuing)"; }
python3 tools/pmc_summary.py $O/mfma_pmc > $O/mfma_pmc_summary.txt 2>&1; cat $O/mfma_pmc_summary.txt
run() {
  cfg=$1; if [ "$cfg" = base ]; then envs="A=1"; else envs="$cfg"; fi
  r=$(env $envs timeout -k 10 600 examples/bin/bench_ceed -bp $2 -p $3 -n $4 -reps 50 -mi 30 2>>$O/err.log | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        r=json.loads(l); print(r['phase'], round(r['ms_per_apply'],4) if 'ms_per_apply' in r else round(r['solve_s']/r['iterations']*1e3,4), round(r['hbm_frac'],4), end='  ')
")
  echo "$cfg bp$2 p$3 n$4: $r" | tee -a $O/summary.txt
}
for p in 2 4 5 6 8; do for cfg in base MFEM_AMD_SCATTER_XCD=1 MFEM_AMD_APPLY_XCD=1 base; do run $cfg 3 $p $((216/p)) || exit 1; done; done
for cfg in base MFEM_AMD_SCATTER_XCD=1; do run $cfg 1 4 54 || exit 1; done
timeout -k 10 900 python -u -m pytest tests/test_gpu_cxx.py -x -q -k full_size --timeout 800 --timeout-method thread > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
