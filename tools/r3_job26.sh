#!/bin/bash
# Set-up time of the C++ parallel layer per rank at equal work per rank (p = 4, ~157k elements = 1e7 dofs a rank), 1 / 2 / 4 ranks
# sharing the one GPU over the host-staged TCP transport (set-up is host work): serial mesh, the rank's part, the space + shared tables.
set -o pipefail
run() { # nranks N
  port=$((32000 + $1*13))
  pids=()
  for r in $(seq 0 $(($1-1))); do
    RANK=$r WORLD_SIZE=$1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$port MFEM_AMD_TRANSPORT=tcp timeout -k 10 500 examples/bin/bench_ceed_p -bp 3 -p 4 -n $2 -reps 3 -mi 5 -rtol 0 > gpurun_out/j26_r$r.log 2>&1 &
    pids+=($!)
  done
  for pid in "${pids[@]}"; do wait $pid || { echo "rank failed"; tail -3 gpurun_out/j26_r*.log; return 1; }; done
  grep -a '"phase"' gpurun_out/j26_r0.log
}
run 1 54 || exit 1
run 2 68 || exit 1
run 4 86 || exit 1
