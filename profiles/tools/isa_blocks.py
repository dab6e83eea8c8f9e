#!/usr/bin/env python3
"""Per-basic-block summary of one kernel in a -save-temps .s file: scratch (spill) ops, MFMAs, global loads / stores,
LDS ops, VALU.  usage: isa_blocks.py file.s kernel_substring [min_instructions]"""
import re
import sys

s = open(sys.argv[1]).read()
i = s.index(sys.argv[2] + '')
i = s.index('\n', s.index(sys.argv[2], s.index('.globl') if False else 0))
k = s[s.index(sys.argv[2]):]
k = k[:k.index('.end_amdhsa_kernel')] if '.end_amdhsa_kernel' in k else k
k = k[:k.index('s_endpgm') + 8] if 's_endpgm' in k and False else k
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 40
stats = [['entry', 0, 0, 0, 0, 0, 0, 0]]
for l in k.split('\n'):
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        stats.append([m.group(1), 0, 0, 0, 0, 0, 0, 0])
        continue
    t = l.strip()
    if not t or t.startswith(('.', ';')):
        continue
    st = stats[-1]
    st[7] += 1
    if 'scratch_' in t: st[1] += 1
    elif 'v_mfma' in t: st[2] += 1
    elif 'global_load' in t: st[3] += 1
    elif 'global_store' in t: st[4] += 1
    elif t.startswith('ds_'): st[5] += 1
    elif t.startswith('v_'): st[6] += 1
print('block            scratch mfma gload gstore lds valu total')
tot = [0] * 7
for st in stats:
    for q in range(7): tot[q] += st[q + 1]
    if st[7] >= thr:
        print('%-16s %6d %5d %5d %6d %4d %5d %5d' % tuple(st))
print('%-16s %6d %5d %5d %6d %4d %5d %5d' % tuple(['TOTAL'] + tot))
