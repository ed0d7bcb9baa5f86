#!/bin/bash
# usage: tools/pmc_bench.sh <tag> <key> <bench.py args...>
# HBM-traffic PMC passes over bench.py itself, one rocprofv3 run per pass (never combined
# with tracing), as MI355X_MICROARCH.md's HBM section prescribes: FETCH_SIZE and WRITE_SIZE
# in separate passes, plus the raw memory-side request counters they derive from.
# <key> ("bp3_p4_n54"): the workload entry written to bench_traffic.json (copied to gpurun_out/ for the way back).
tag=$1; key=$2; shift; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/pmcb_$tag
mkdir -p $out
i=0
for ctrs in "FETCH_SIZE" "WRITE_SIZE" \
            "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" \
            "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  timeout -s KILL 150 rocprofv3 --pmc $ctrs --output-format csv -d $out/pass$i -- python3 bench.py --no-cpu --no-cxx --no-orders --no-sizes --cg-iters 0 --steps 6 --warmup 2 "$@" > $out/pass$i.log 2>&1 || { echo "pass $i failed"; tail -3 $out/pass$i.log; exit 1; }
done
python3 tools/pmc_summary.py $out --traffic-json $key "profiles/r3/pmc_bench_$tag.txt (tools/pmc_bench.sh $tag $key $*)" && cp bench_traffic.json gpurun_out/bench_traffic.json
