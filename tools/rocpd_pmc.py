#!/usr/bin/env python3
"""Print per-kernel PMC sums/averages from a rocprofv3 rocpd sqlite db: python tools/rocpd_pmc.py x_results.db"""
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info('counters_collection')")]
rows = db.execute("select * from counters_collection").fetchall()
ci = {c: i for i, c in enumerate(cols)}
name_col = "kernel_name" if "kernel_name" in ci else ("name" if "name" in ci else None)
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    k = (str(r[ci[name_col]]).replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:60], r[ci["counter_name"]])
    agg[k][0] += 1; agg[k][1] += float(r[ci["value"]])
flt = sys.argv[2] if len(sys.argv) > 2 else None       # optional kernel-name substring
for (kn, cn), (n, v) in sorted(agg.items()):
    if flt and flt not in kn:
        continue
    print(f"{kn:62s} {cn:28s} n={n:4d} avg={v/n:16.1f}")
