#!/usr/bin/env python3
"""GPU busy fraction of a bench.py run from a `rocprofv3 --kernel-trace --output-format csv` trace: the share of the steady-state
window (the middle of the timed steps) in which at least one kernel is executing, the gaps, and how much kernel time overlaps.
usage: python tools/gpu_busy.py <dir with *kernel_trace.csv> [lo_frac hi_frac]"""
import csv
import glob
import json
import os
import sys

import numpy as np


def main():
    d = sys.argv[1]
    lo_f = float(sys.argv[2]) if len(sys.argv) > 2 else 0.6
    hi_f = float(sys.argv[3]) if len(sys.argv) > 3 else 0.9
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(open(f)))
    t0, t1 = rows[0][0], max(e for _, e in rows)
    lo, hi = t0 + lo_f * (t1 - t0), t0 + hi_f * (t1 - t0)
    iv = sorted((max(s, lo), min(e, hi)) for s, e in rows if e > lo and s < hi)
    cov, gaps = 0.0, []
    cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce:
            cov += ce - cs
            gaps.append(s - ce)
            cs, ce = s, e
        else:
            ce = max(ce, e)
    cov += ce - cs
    ksum = sum(e - s for s, e in iv)
    g = np.array(gaps or [0])
    print(json.dumps({"window_ms": round((hi - lo) / 1e6, 2), "busy_frac": round(cov / (hi - lo), 4),
                      "kernel_time_sum_over_busy_time": round(ksum / cov, 3), "kernels": len(iv),
                      "gaps": {"n": len(gaps), "sum_ms": round(float(g.sum()) / 1e6, 3), "mean_us": round(float(g.mean()) / 1e3, 2),
                               "p90_us": round(float(np.percentile(g, 90)) / 1e3, 2), "max_us": round(float(g.max()) / 1e3, 1)}}))


if __name__ == "__main__":
    main()
