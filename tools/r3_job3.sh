#!/bin/bash
# round 3, job 3: traffic counters and kernel stats of the packed apply; direct-row policy at low order and at 1e8 dofs
set -o pipefail
O=gpurun_out/r3job3; mkdir -p $O
bash tools/pmc_bench.sh r3p4 bp3_p4_n54 > $O/pmc_bp3.txt 2>&1 || { tail -5 $O/pmc_bp3.txt; exit 1; }
bash tools/pmc_bench.sh r3bp1p4 bp1_p4_n54 --problem bp1 > $O/pmc_bp1.txt 2>&1 || { tail -5 $O/pmc_bp1.txt; exit 1; }
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --no-cpu --no-cxx --no-orders --no-sizes > $O/bench_stats.json 2>$O/bench_stats.err || { tail -5 $O/bench_stats.err; exit 1; }
find $O/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/bench_p4_kernel_stats.csv
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
for bp in 3 1; do for p in 2 3; do for cfg in MFEM_AMD_DIRECT=0 MFEM_AMD_DIRECT=2; do run $cfg $bp $p $((216/p)) || exit 1; done; done; done
for bp in 3 1; do for cfg in MFEM_AMD_DIRECT=0 MFEM_AMD_DIRECT=2; do run $cfg $bp 4 25 || exit 1; done; done
for bp in 3 1; do for cfg in MFEM_AMD_DIRECT=0 MFEM_AMD_DIRECT=2; do run $cfg $bp 4 116 || exit 1; done; done
timeout -k 10 900 python -u -m pytest tests/test_gpu_cxx.py -x -q -k full_size --timeout 800 --timeout-method thread > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
