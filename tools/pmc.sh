#!/bin/bash
# usage: tools/pmc.sh <tag> <kbench args...>  -- runs the PMC passes one by one (never combined with tracing)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmc_$tag
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum GRBM_GUI_ACTIVE" \
            "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
            "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d gpurun_out/pmc_$tag/pass$i -- python3 tools/kbench.py "$@" > gpurun_out/pmc_$tag/pass$i.log 2>&1 || { echo "pass $i failed"; tail -3 gpurun_out/pmc_$tag/pass$i.log; exit 1; }
done
python3 tools/pmc_summary.py gpurun_out/pmc_$tag
