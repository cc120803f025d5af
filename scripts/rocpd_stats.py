#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) as a per-kernel stats table (CSV on stdout).
usage: rocpd_stats.py trace_results.db [min_start_fraction]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, start, end from kernels").fetchall()
agg = {}
for name, s, e in rows:
    a = agg.setdefault(name, [0, 0, 1 << 62, 0])
    d = e - s
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
tot = sum(a[1] for a in agg.values()) or 1
print("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs")
for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"\"{name}\",{a[0]},{a[1]},{a[1] / a[0]:.1f},{100.0 * a[1] / tot:.2f},{a[2]},{a[3]}")
