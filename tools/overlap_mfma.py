#!/usr/bin/env python3
"""In-situ MFMA fraction of the BENCHED (overlapped) schedule from a `rocprofv3 --kernel-trace` run of `bench.py --lean` (VERDICT r02
"What's weak" 7): `roofline.frac` is measured launch by launch on one stream; the headline runs ~20 streams of two codec objects.  Here:
inside the steady-state window of the timed steps, FLOPs of the conv-family launches that lie in the window divided by (a) the time at
least one conv-family kernel is executing (union of their intervals) and (b) the window itself.
FLOPs: the conv family makes LAUNCHES_PER_STEP launches and GFLOP_PER_STEP algorithmic GFLOP per step (bench.py's roofline leg,
`launches_per_step` / `algorithmic_gflop_per_step`); a window over several steps holds launches in the per-step mix.
usage: python tools/overlap_mfma.py <dir with *_results.db> <bench json of that run> [lo_frac hi_frac] > profiles/rNN_overlap_schedule_mfma.json"""
import glob
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PEAK = 157.3


def union(iv):
    iv = sorted(iv)
    cov, (cs, ce) = 0, iv[0]
    for s, e in iv[1:]:
        if s > ce:
            cov += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return cov + ce - cs


def main():
    d, bj = sys.argv[1], json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    lo_f = float(sys.argv[3]) if len(sys.argv) > 3 else 0.45
    hi_f = float(sys.argv[4]) if len(sys.argv) > 4 else 0.85
    db = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    lo, hi = t0 + lo_f * (t1 - t0), t0 + hi_f * (t1 - t0)
    conv = [(s, e) for n, s, e in rows if "conv_igemm" in n and s >= lo and e <= hi]
    allk = [(max(s, lo), min(e, hi)) for n, s, e in rows if e > lo and s < hi]
    rf = bj["roofline"]
    gflop = len(conv) / rf["launches_per_step"] * rf["algorithmic_gflop_per_step"]
    busy_conv, busy_all, win = union(conv), union(allk), hi - lo
    from bench import source_hash
    print(json.dumps({"source_hash": source_hash(),
                      "source": "rocprofv3 --kernel-trace of `python3 bench.py --steps 30 --warmup 2 --lean` (default overlapped schedule), folded by tools/overlap_mfma.py",
                      "window_ms": round(win / 1e6, 2), "window_frac_of_trace": [lo_f, hi_f], "steps_in_window": round(len(conv) / rf["launches_per_step"], 2),
                      "conv_launches_in_window": len(conv), "conv_busy_ms": round(busy_conv / 1e6, 2), "any_kernel_busy_ms": round(busy_all / 1e6, 2),
                      "conv_kernel_time_sum_over_conv_busy": round(sum(e - s for s, e in conv) / busy_conv, 3),
                      "tflops_over_conv_busy_time": round(gflop / (busy_conv / 1e9) / 1e3, 2), "frac_of_f32_mfma_peak_over_conv_busy_time": round(gflop / (busy_conv / 1e9) / 1e3 / PEAK, 4),
                      "tflops_over_window": round(gflop / (win / 1e9) / 1e3, 2), "frac_of_f32_mfma_peak_over_window": round(gflop / (win / 1e9) / 1e3 / PEAK, 4),
                      "bench_value_mp_s_under_profiler": bj["value"]}))


if __name__ == "__main__":
    main()
