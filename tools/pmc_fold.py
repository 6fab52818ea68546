#!/usr/bin/env python3
"""Average every collected PMC counter per kernel (and grid size) out of rocprofv3's rocpd output.
usage: python tools/pmc_fold.py <dir with *_results.db> [kernel substring]"""
import collections
import glob
import os
import sqlite3
import sys

db = glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True)[0]
sub = sys.argv[2] if len(sys.argv) > 2 else "conv_igemm"
cur = sqlite3.connect(db).cursor()
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for name, grid, cname, val, dur in cur.execute("select kernel_name, grid_size, counter_name, value, duration from counters_collection"):
    if sub not in name:
        continue
    k = (name.split("(")[0][-48:], grid)
    a = agg[k][cname]
    a[0] += 1
    a[1] += float(val)
    d = agg[k]["__duration_ns"]
    d[0] += 1
    d[1] += float(dur)
for k, cs in agg.items():
    print(k)
    for c, (n, v) in sorted(cs.items()):
        print(f"   {c:32s} {v / n:16.1f}   (n={n})")
