#!/usr/bin/env python3
"""Per-stream view of a rocprofv3 rocpd kernel trace: for a window of the launch sequence, every sgs kernel with its queue /
stream, start, duration, and the gap since the previous kernel of the same stream ended.
usage: rocpd_chain.py trace_results.db [first_fraction [count]]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
ktab = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel" in t and "dispatch" in t][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({ktab})")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
qcol = next((c for c in ("stream_id", "stream", "queue_id", "queue") if c in cols), None)
sel = f"select {name_col}, start, end" + (f", {qcol}" if qcol else ", 0") + f" from {ktab}"
rows = [r for r in cur.execute(sel) if "sgs::" in r[0] or "rocclr" in r[0]]
rows.sort(key=lambda r: r[1])
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 80
rows = rows[int(len(rows) * f0):][:cnt]
t0 = rows[0][1]; last = {}
print(f"columns: {cols}")
for n, s, e, q in rows:
    gap = (s - last[q]) / 1e3 if q in last else float("nan")
    last[q] = e
    short = n.split("(")[0].replace("void ", "").replace("sgs::", "").replace("__amd_rocclr_", "rocclr:")
    print(f"q{q}  start {(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {gap:7.1f}  {short}")
