#!/usr/bin/env python3
"""The launch sequence of a rocprofv3 rocpd database (kernel trace) as text: start, end, duration, queue / stream, grid, kernel — for a window
of the sgs launches.  usage: rocpd_dump.py trace_results.db [first_fraction last_fraction]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
ktab = "kernels" if "kernels" in tabs else [t for t in tabs if "kernel" in t and "dispatch" in t][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({ktab})")]
print("# table", ktab, "columns", cols)
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
extra = [c for c in cols if c in ("queue_id", "stream_id", "queue", "stream", "grid_x", "grid_size_x", "grid_size", "grid_y", "grid_size_y", "workgroup_x", "tid")]
rows = [r for r in cur.execute(f"select {name_col}, start, end, {', '.join(extra) if extra else '0'} from {ktab}") if "sgs::" in r[0]]
rows.sort(key=lambda r: r[1])
f0 = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
f1 = float(sys.argv[3]) if len(sys.argv) > 3 else 0.5
rows = rows[int(len(rows) * f0):int(len(rows) * f1)]
t0 = rows[0][1]
print("# start_us end_us dur_us", extra, "kernel")
for r in rows:
    n = r[0].split("(")[0].replace("void sgs::", "").replace("sgs::", "")
    print(f"{(r[1] - t0) / 1e3:10.1f} {(r[2] - t0) / 1e3:10.1f} {(r[2] - r[1]) / 1e3:8.1f}  {' '.join(str(v) for v in r[3:])}  {n}")
