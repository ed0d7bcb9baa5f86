#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmc_calib; mkdir -p $out
timeout -k 5 60 rocprofv3 -L > $out/counters.txt 2>&1
grep -o "TCC_EA0_RDREQ[A-Z0-9_]*\|TCC_EA0_WRREQ[A-Z0-9_]*\|TCC_MISS[A-Za-z_]*\|TCC_HIT[A-Za-z_]*\|TCC_REQ[A-Za-z_]*\|TCC_READ[A-Za-z_]*" $out/counters.txt | sort -u > $out/tcc_names.txt
cat $out/tcc_names.txt | tr '\n' ' '; echo
i=0
for ctrs in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum TCC_HIT_sum" "FETCH_SIZE" "WRITE_SIZE" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_REQ_sum TCC_READ_sum"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --pmc $ctrs --output-format csv -d $out/pass$i -- python3 tools/pmc_calib.py > $out/pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out/pass$i.log; exit 1; }
done
python3 tools/pmc_summary.py $out
