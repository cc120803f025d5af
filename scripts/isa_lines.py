#!/usr/bin/env python3
"""VALU / SALU / LDS / VMEM instructions of one kernel per SOURCE LINE (build with -gline-tables-only -save-temps):
usage: isa_lines.py file.s kernel_substring [lo hi]   — prints lines of sgs_kernels.h in [lo, hi] with their static counts"""
import re, sys
from collections import defaultdict
f, kern = sys.argv[1], sys.argv[2]
lo, hi = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (0, 10**9)
lines = open(f).read().split('\n')
start = next(n for n, l in enumerate(lines) if l.startswith('_ZN') and kern in l and l.rstrip().endswith(':') or (l.startswith('_ZN') and kern in l and ':' in l and not l.startswith('\t')))
end = next(n for n in range(start, len(lines)) if lines[n].startswith('.Lfunc_end'))
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2))
cur = (None, 0)
cnt = defaultdict(lambda: [0, 0, 0, 0])
for l in lines[start:end]:
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m: cur = (files.get(int(m.group(1)), '?'), int(m.group(2))); continue
    t = l.strip()
    if not t or t[0] in '.;' or t.endswith(':'): continue
    op = t.split()[0]
    k = 0 if op.startswith('v_') else 1 if op.startswith('s_') else 2 if op.startswith('ds_') else 3 if op.startswith(('global', 'buffer', 'flat', 'scratch')) else None
    if k is not None: cnt[cur][k] += 1
tot = [0, 0, 0, 0]
for (fn, ln), c in sorted(cnt.items(), key=lambda kv: (str(kv[0][0]), kv[0][1])):
    for i in range(4): tot[i] += c[i]
    if fn and fn.endswith('sgs_kernels.h') and lo <= ln <= hi:
        print(f"{ln:5d}  VALU {c[0]:4d} SALU {c[1]:4d} LDS {c[2]:3d} VMEM {c[3]:3d}")
print("total", tot)
