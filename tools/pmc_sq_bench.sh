#!/bin/bash
# usage: tools/pmc_sq_bench.sh <tag> <library file under mfem_amd/> <bench.py args...>
# SQ wave-state counters of the apply kernels under bench.py itself, one rocprofv3 --pmc run per pass (never combined with tracing).
tag=$1; lib=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MFEM_AMD_LIB=$GRAFT_REPO_ROOT/mfem_amd/$lib
out=gpurun_out/pmcsq_$tag
mkdir -p $out
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
            "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $ctrs --output-format csv -d $out/pass$i -- python3 bench.py --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 0 --steps 6 --warmup 2 "$@" > $out/pass$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/pass$i.log; exit 1; }
done
python3 tools/pmc_summary.py $out | grep -A 20 "k_pa_apply<" | head -24
