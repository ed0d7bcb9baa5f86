#!/bin/bash
# usage: [BPS="2 4 6"] tools/sweep_bp.sh <tag> [max_exp]   -- BP1/BP3/BP5 (or $BPS), p = 1..8, ~1e5..1e<max_exp> dofs through the
# C++ drivers (examples/bin/bench_ceed); one JSON line per (case, phase) into gpurun_out/sweep_<tag>.jsonl
tag=$1; maxe=${2:-7}
out=gpurun_out/sweep_$tag.jsonl; : > $out
for bp in ${BPS:-3 1 5}; do for e in $(seq 5 $maxe); do for p in 1 2 3 4 5 6 7 8; do
  n=$(python3 -c "print(max(1, round((10**$e)**(1/3)/$p)))")
  timeout -k 10 300 examples/bin/bench_ceed -bp $bp -p $p -n $n -reps 20 -mi 50 -rtol 0 | grep '^{' >> $out || { echo "failed bp$bp p$p n$n"; exit 1; }
done; done; echo "bp$bp done"; done
python3 - $out <<'PY'
import json,sys
rows=[json.loads(l) for l in open(sys.argv[1])]
print("bp p dofs      apply GDOF/s  frac   | PCG GDOF*it/s frac")
k={(r['bp'],r['p'],r['dofs']):r for r in rows if r['phase']=='kernel'}
for r in rows:
    if r['phase']=='problem':
        a=k[(r['bp'],r['p'],r['dofs'])]
        print(f"{r['bp']} {r['p']} {r['dofs']:>10d} {a['MDof/s']/1e3:10.2f} {a['hbm_frac']:6.3f} | {r['MDof_it/s']/1e3:10.2f} {r['hbm_frac']:6.3f}")
PY
