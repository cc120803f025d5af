#!/usr/bin/env python3
"""Average PMC counters per kernel from rocprofv3 counter_collection CSVs: pmc_summary.py dir [skip_first_launches]"""
import csv, sys, os, collections, json
d = sys.argv[1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
out = collections.defaultdict(dict)
for f in sorted(os.listdir(d)):
    if not f.endswith(".csv"): continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(os.path.join(d, f)) as fh:
        for row in csv.DictReader(fh):
            k = row.get("Kernel_Name", "")
            k = k.split("(")[0].replace("void ", "")
            agg[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, cs in agg.items():
        for c, v in cs.items():
            v = v[skip:] if len(v) > skip else v
            out[k][c] = sum(v) / len(v)
            out[k]["_launches"] = len(v)
print(json.dumps(out, indent=1))
