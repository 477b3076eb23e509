#!/usr/bin/env python3
"""Count instruction classes per basic block of one kernel in a hipcc -S --cuda-device-only listing.
   tools/isa_count.py file.s KERNEL_SUBSTRING [--blocks]
Vector-ALU instructions (and where they sit) are what bounds the lattice kernels: DESIGN.md section 5.7."""
import re, sys
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*:', l) and key in l)
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith('s_endpgm'))
blocks, cur = [], ['entry', []]
for l in lines[start + 1:end + 1]:
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m:
        blocks.append(cur); cur = [m.group(1), []]; continue
    s = l.strip()
    if not s or s[0] in ';.': continue
    cur[1].append(s.split()[0])
blocks.append(cur)
tot = {}
for name, ins in blocks:
    c = dict(valu=0, mfma=0, trans=0, ds=0, vmem=0, salu=0, br=0)
    for i in ins:
        if i.startswith('v_mfma'): c['mfma'] += 1
        elif i.startswith('v_'):
            c['valu'] += 1
            if re.match(r'v_(sin|cos|rsq|sqrt|rcp|exp|log)', i): c['trans'] += 1
        elif i.startswith('ds_'): c['ds'] += 1
        elif i.startswith(('global_', 'buffer_', 'scratch_', 'flat_')): c['vmem'] += 1
        elif i.startswith(('s_cbranch', 's_branch')): c['br'] += 1
        elif i.startswith('s_'): c['salu'] += 1
    for k, v in c.items(): tot[k] = tot.get(k, 0) + v
    if '--blocks' in sys.argv: print(name, len(ins), c)
print('static totals', tot)
