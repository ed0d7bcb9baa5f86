#!/bin/bash
# BP1 and BP5 at p = 4: measured HBM traffic of the two apply kernels (bench_traffic.json) and the SQ wait/busy
# counters of the BP1 element kernel (is it latency-bound or at the ceiling of its real traffic?)


cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/sq_bp1; mkdir -p $out
i=0
for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "GRBM_GUI_ACTIVE TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $ctrs --output-format csv -d $out/pass$i -- python3 tools/kbench.py --problem bp1,bp3 --p 4 --which l2e --steps 5 > $out/pass$i.log 2>&1 || { echo "sq pass $i failed"; tail -3 $out/pass$i.log; exit 1; }
done
python3 tools/pmc_summary.py $out > /dev/null && grep -A30 "_apply<" $out/summary.txt
