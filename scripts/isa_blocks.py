#!/usr/bin/env python3
"""Instruction mix of the basic blocks of one kernel that contain a given opcode (default v_exp_f32).
usage: isa_blocks.py file.s kernel_substring [opcode] [--dump]"""
import re, sys
from collections import Counter
f, kern = sys.argv[1], sys.argv[2]
op = sys.argv[3] if len(sys.argv) > 3 and not sys.argv[3].startswith('--') else 'v_exp_f32'
lines = open(f).read().split('\n')
start = next(n for n, l in enumerate(lines) if l.startswith('_ZN') and kern in l and l.rstrip().split(':')[0].endswith(l.split(':')[0]))
end = next(n for n in range(start, len(lines)) if lines[n].startswith('.Lfunc_end'))
idx = [n for n in range(start, end) if op in lines[n]]
def bounds(n):
    a = n
    while a > start and not re.match(r'\.LBB\d+_\d+:', lines[a]): a -= 1
    b = n
    while b < end - 1 and not re.match(r'\.LBB\d+_\d+:', lines[b + 1]): b += 1
    return a, b
seen = set()
for n in idx:
    a, b = bounds(n)
    if (a, b) in seen: continue
    seen.add((a, b))
    ins = [l.strip().split()[0] for l in lines[a + 1:b + 1] if l.strip() and not l.strip().startswith((';', '.'))]
    def cls(x):
        if x.startswith('v_'): return 'VALU'
        if x.startswith(('s_waitcnt', 's_cbranch', 's_nop', 's_branch', 's_barrier')): return x
        if x.startswith('s_'): return 'SALU'
        if x.startswith('ds_'): return 'LDS'
        if x.startswith(('scratch', 'buffer', 'global', 'flat')): return 'VMEM'
        return x
    print(lines[a], f'{op} x', sum(1 for x in ins if x == op), dict(Counter(cls(x) for x in ins)))
    print('   ', sorted(Counter(x for x in ins if x.startswith('v_')).items(), key=lambda kv: -kv[1]))
    if '--dump' in sys.argv:
        print('\n'.join(lines[a:b + 1]))
