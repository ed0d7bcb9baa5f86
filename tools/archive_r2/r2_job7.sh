#!/bin/bash
# the placement / history dependent slowdown: reproduce under the kernel trace
set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/slow_trace -o seq -- python3 tools/kbench.py --problem bp3,bp1 --p 1,3,5,7,8 --which l2e > gpurun_out/slow_seq.txt 2>&1
timeout -k 10 300 python3 -u tools/kbench.py --problem bp1 --p 8 --which l2e > gpurun_out/slow_alone.txt 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/slow_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print(rows[0].keys())
by = collections.defaultdict(list)
for r in rows:
    by[r['Kernel_Name'][:70]].append(r)
for k, v in by.items():
    if 'k_pa_apply' not in k: continue
    d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp']))/1e3 for r in v]
    gaps = [(int(v[i+1]['Start_Timestamp']) - int(v[i]['End_Timestamp']))/1e3 for i in range(len(v)-1)]
    extra = {c: v[0].get(c) for c in v[0] if 'cratch' in c or 'Private' in c or 'VGPR' in c or 'LDS' in c}
    print(k, 'n=%d' % len(v), 'dur us min/med/max = %.1f %.1f %.1f' % (min(d), sorted(d)[len(d)//2], max(d)),
          'gap med = %.1f' % (sorted(gaps)[len(gaps)//2] if gaps else 0), extra)
PY
