#!/bin/bash
# (1) whole apply at 1e7 dofs eager vs replayed from a HIP graph; (2) SQ counters of the p = 5, 6 element kernels
mkdir -p gpurun_out
for v in 0 1 0 1; do
  MFEM_AMD_BENCH_GRAPH=$v timeout -k 10 300 python3 bench.py --no-cpu --no-sizes --cg-iters 0 > gpurun_out/bench_graph.log 2>&1 || { tail -3 gpurun_out/bench_graph.log; exit 1; }
  grep -a "^{" gpurun_out/bench_graph.log | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline())
print('GRAPH=$v by_order us:', ' '.join('%.1f' % (r['ms_per_step']*1e3) for r in d['by_order']), d['by_order'][3]['launch'])"
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/sq_p56; rm -rf $out; mkdir -p $out
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $ctrs --output-format csv -d $out/pass$i -- python3 tools/kbench.py --problem bp3 --p 5,6,8 --which l2e --steps 5 > $out/pass$i.log 2>&1 || { echo "sq pass $i failed"; tail -3 $out/pass$i.log; exit 1; }
done
python3 tools/pmc_summary.py $out > /dev/null && grep -A20 "apply<" $out/summary.txt
