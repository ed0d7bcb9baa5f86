#!/usr/bin/env python3
"""Per-kernel resources of a hipcc -save-temps assembly file (tuning aid): template arguments of k_pa_apply
<KIND,D,Q,NBZ,GATHER,ACCUM,BLK,ABL,PERSIST,PF,RB,VEC,NTP>, SGPRs, VGPRs, scratch bytes per lane, LDS."""
import re, sys
s = open(sys.argv[1]).read()
for m in re.finditer(r'\.set (_ZN3mfa\w+)\.private_seg_size, (\d+)', s):
    name, scr = m.group(1), int(m.group(2))
    args = re.findall(r'L[ib](\d+)E', name)
    blk = s[m.end():m.end() + 3000]
    g = lambda k: (re.search(k + r': (\d+)', blk) or [None, '?'])[1]
    if len(sys.argv) > 2 and sys.argv[2] == 'scratch' and scr == 0:
        continue
    print(('pa_apply<' + ','.join(args[:13]) + '>') if 'k_pa_apply' in name else name[:60],
          'sgpr', g('TotalNumSgprs'), 'vgpr', g('NumVgprs'), 'scratch', scr, 'lds', g('LDSByteSize'))
