#!/usr/bin/env python3
"""Instruction mix and register use of one kernel in a hipcc --save-temps .s file.
usage: asm_stats.py file.s kernel-name-substring"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
pat = sys.argv[2]
names = [m for m in re.findall(r'^(\S+):\s*;\s*@', s, re.M) if pat in m]
for name in names:
    i = s.index('\n' + name + ':')
    j = s.index('.Lfunc_end', i)
    body = s[i:j]
    ins = [l.split()[0] for l in body.splitlines() if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    print(name, len(ins), 'instructions')
    for k, v in Counter(ins).most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 25):
        print('  %-28s %d' % (k, v))
    k = s.index('.amdhsa_kernel ' + name)
    blk = s[k:k + 4000]
    for key in ('next_free_vgpr', 'next_free_sgpr', 'private_segment_fixed_size', 'accum_offset', 'group_segment_fixed_size'):
        m = re.search(key + r'\s+(\S+)', blk)
        print('  ', key, m.group(1) if m else None)
