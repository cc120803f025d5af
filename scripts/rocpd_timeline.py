#!/usr/bin/env python3
"""Concurrency summary of a rocprofv3 rocpd database (kernel trace): over the span of the sgs kernels, the share
of time with 0, 1, 2, ... kernels running, and per-kernel average duration.
usage: rocpd_timeline.py trace_results.db [first_fraction last_fraction]   (window of the launch sequence)"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
ktab = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel" in t and "dispatch" in t][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({ktab})")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = [(n, s, e) for n, s, e in cur.execute(f"select {name_col}, start, end from {ktab}") if "sgs::" in n]
rows.sort(key=lambda r: r[1])
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
rows = rows[int(len(rows) * f0):int(len(rows) * f1)]    # bench.py: warm-up, timed region, then the one-at-a-time post-pass
t0, t1 = rows[0][1], max(r[2] for r in rows)
ev = []
for n, s, e in rows:
    ev.append((s, 1)); ev.append((e, -1))
ev.sort()
hist, cur_n, last = {}, 0, t0
for t, d in ev:
    hist[cur_n] = hist.get(cur_n, 0) + (t - last)
    cur_n += d; last = t
span = t1 - t0
print(f"span {span / 1e6:.2f} ms, {len(rows)} launches")
for k in sorted(hist):
    print(f"  {k} kernels running: {100.0 * hist[k] / span:5.1f} %")
agg = {}
for n, s, e in rows:
    a = agg.setdefault(n.split('(')[0], [0, 0]); a[0] += 1; a[1] += e - s
for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"  {n:40s} calls {a[0]:5d}  avg {a[1] / a[0] / 1e3:8.1f} us  sum/span {a[1] / span:5.2f}")
