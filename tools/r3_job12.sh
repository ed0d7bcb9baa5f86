#!/bin/bash
# SQ / TCP / TCC counters of the two apply kernels in context (bench.py, BP3 p = 4, reference numbering): what the transpose-gather waits for.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_e2l_ctx
mkdir -p $out
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" \
            "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
            "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_TCP_STATE_READ_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
            "TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $ctrs --output-format csv -d $out/pass$i -- python3 bench.py --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 0 --steps 6 --warmup 2 > $out/pass$i.log 2>&1 || { echo "pass $i failed ($ctrs)"; tail -3 $out/pass$i.log; }
done
python3 tools/pmc_summary.py $out | grep -A40 -E "k_scatter|k_pa_apply" 
