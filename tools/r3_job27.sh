#!/bin/bash
# the same 4-rank run with the previous build of the parallel layer (global space on every rank) and the present one: wall time and peak RSS of rank 0
set -o pipefail
run() { # tag exe libdir
  port=$((33000 + RANDOM % 500))
  pids=()
  for r in 0 1 2 3; do
    if [ $r = 0 ]; then
      LD_LIBRARY_PATH=$3:$PWD/mfem_amd:$LD_LIBRARY_PATH RANK=$r WORLD_SIZE=4 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$port MFEM_AMD_TRANSPORT=tcp timeout -k 10 500 python3 -c "
import resource, subprocess, sys, time
t = time.time(); rc = subprocess.call(sys.argv[1:]); dt = time.time() - t
print('rank 0: wall %.2f s, peak RSS %.0f MB' % (dt, resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss/1024.0)); sys.exit(rc)" $2 -bp 3 -p 4 -n 86 -reps 3 -mi 5 -rtol 0 > gpurun_out/j27_$1_r$r.log 2>&1 &
    else
      LD_LIBRARY_PATH=$3:$PWD/mfem_amd:$LD_LIBRARY_PATH RANK=$r WORLD_SIZE=4 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=$port MFEM_AMD_TRANSPORT=tcp timeout -k 10 500 $2 -bp 3 -p 4 -n 86 -reps 3 -mi 5 -rtol 0 > gpurun_out/j27_$1_r$r.log 2>&1 &
    fi
    pids+=($!)
  done
  for pid in "${pids[@]}"; do wait $pid || { echo "rank failed"; head -5 gpurun_out/j27_$1_r0.log gpurun_out/j27_$1_r1.log; return 1; }; done
  echo "== $1"; grep -a -E '"phase"|rank 0' gpurun_out/j27_$1_r0.log
}
run old tools/tmp_old/bench_ceed_p_old $PWD/tools/tmp_old || exit 1
run new examples/bin/bench_ceed_p $PWD/mfem_amd || exit 1
