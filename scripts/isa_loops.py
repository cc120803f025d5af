#!/usr/bin/env python3
"""Loops (backward branches) of one kernel in an ISA dump, with their instruction mix, innermost first.
usage: isa_loops.py file.s kernel_substring [min_valu]"""
import re, sys
from collections import Counter
f, kern = sys.argv[1], sys.argv[2]
minv = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lines = open(f).read().split('\n')
start = next(n for n, l in enumerate(lines) if l.startswith('_ZN') and kern in l)
end = next(n for n in range(start, len(lines)) if lines[n].startswith('.Lfunc_end'))
labels = {}
for n in range(start, end):
    m = re.match(r'(\.LBB\d+_\d+):', lines[n])
    if m: labels[m.group(1)] = n
loops = []
for n in range(start, end):
    m = re.match(r'\s+s_cbranch_\w+\s+(\.LBB\d+_\d+)', lines[n]) or re.match(r'\s+s_branch\s+(\.LBB\d+_\d+)', lines[n])
    if m and m.group(1) in labels and labels[m.group(1)] < n:
        loops.append((labels[m.group(1)], n))
def mix(a, b):
    ins = [l.strip().split()[0] for l in lines[a + 1:b + 1] if l.strip() and not l.strip().startswith((';', '.'))]
    c = Counter()
    for x in ins:
        if x.startswith('v_'): c['VALU'] += 1
        elif x.startswith('ds_'): c['LDS'] += 1
        elif x.startswith(('global', 'buffer', 'flat', 'scratch')): c['VMEM'] += 1
        elif x.startswith('s_waitcnt'): c['wait'] += 1
        elif x.startswith('s_barrier'): c['barrier'] += 1
        elif x.startswith('s_cbranch') or x.startswith('s_branch'): c['br'] += 1
        elif x.startswith('s_'): c['SALU'] += 1
    return c, Counter(x for x in ins if x.startswith('v_'))
for a, b in sorted(set(loops), key=lambda ab: ab[1] - ab[0]):
    c, v = mix(a, b)
    if c['VALU'] < minv: continue
    inner = sum(1 for (x, y) in set(loops) if x >= a and y <= b) - 1
    print(f"{lines[a].split(':')[0]:14s} lines {a - start:5d}..{b - start:5d}  inner loops {inner:2d}  {dict(c)}")
    print('      ', v.most_common(8))
