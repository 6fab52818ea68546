#!/usr/bin/env python3
"""Kernel summary (the `rocprofv3 --kernel-trace --stats` table) out of rocprofv3's rocpd SQLite output.
usage: python tools/rocpd_stats.py <dir with *_results.db> > profiles/rNN_kernel_stats.csv"""
import glob
import os
import sqlite3
import sys

db = glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True)[0]
cur = sqlite3.connect(db).cursor()
print('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs"')
rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
for name, n, s, a, mn, mx in rows:
    print(f'"{name}",{n},{s},{a:.1f},{100.0 * s / tot:.3f},{mn},{mx}')
