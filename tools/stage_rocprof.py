#!/usr/bin/env python3
"""Fold a `rocprofv3 --kernel-trace --output-format csv` run of tools/stage_bench.py into HBM GB/s per mask / entropy-prep stage
(kernel durations as the profiler saw them -- not HIP events; VERDICT r01 "What's weak" 6).

usage: python tools/stage_rocprof.py <dir with *kernel_trace.csv> <n_images> [<--pmc FETCH_SIZE dir> <--pmc WRITE_SIZE dir>] > profiles/rNN_x_stage_kernels_rocprof.json

With the two PMC directories (separate `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes over the same command) every
stage also carries the bytes the L2's memory side really moved per call (FETCH_SIZE x2 on gfx950, Infinity-Cache hits included) and the
GB/s they make at the trace's duration, next to the algorithmic figure (VERDICT r02 "What's weak" 6).

stage_bench.py calls, in this order: the quantile once, then (1 + 10) times each of: quantile, encoder enhancement prep, encoder base
prep, decoder index, dequantise.  The kernel trace is cut into those calls by start time; a call of the quantile on a Config-4 slice
is three kernels (sample / bracket / final; a one-launch form was built and rejected in round 4, pc_stages.hip); every other call is one kernel;
the per-kernel averages are reported too."""
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def pmc_calls(d, counter):
    """per stage call, in dispatch order: summed counter value (KB) of its kernels -- the same segmentation as the trace's"""
    import sqlite3
    db = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select dispatch_id, kernel_name, value from counters_collection where counter_name = ? order by dispatch_id", (counter,)).fetchall()
    return [(n, float(v)) for _, n, v in rows if any(t in n for t in ("quantile", "gc_prep", "gc_dequant"))]


def segment(ks):
    calls, cur = [], None
    for name, ns in ks:
        zero = any(t in name for t in ("gc_prep_kernel<0>", "gc_prep_kernelILi0", "gc_prep_vec_kernel<0", "gc_prep_vec_kernelILi0"))
        one = any(t in name for t in ("gc_prep_kernel<1>", "gc_prep_kernelILi1", "gc_prep_vec_kernel<1", "gc_prep_vec_kernelILi1"))
        key = "quantile" if "quantile" in name else ("gc_prep_kernel<0>" if zero else ("gc_prep_kernel<1>" if one else "gc_dequant"))
        if key == "quantile":
            if cur is None or cur[0] != "quantile" or "sample" in name or "thr_kernel" in name or "onepass" in name:
                cur = ["quantile", 0, 0]
                calls.append(cur)
            cur[1] += ns; cur[2] += 1
        else:
            cur = [key, ns, 1]
            calls.append(cur)
    return calls


def main():
    d, B = sys.argv[1], int(sys.argv[2])
    pmc = None
    if len(sys.argv) > 4:
        pmc = (segment(pmc_calls(sys.argv[3], "FETCH_SIZE")), segment(pmc_calls(sys.argv[4], "WRITE_SIZE")))
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    ks = [(r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows
          if any(t in r["Kernel_Name"] for t in ("quantile", "gc_prep", "gc_dequant"))]
    n = B * 64 * 64 * 32
    stages = [("quantile threshold (layers/masking.py:218: torch.quantile per image)", "quantile", 4),
              ("gc_prep_kernel<0> enhancement slice (mask + index + quantise + dequantise)", "gc_prep_kernel<0>", 32),   # read {scale, y_enh, y_base, mu} 16 B + write {mask, index, symbol, y_hat} 16 B (rounds 1-2 billed the mask twice: 36)
              ("gc_prep_kernel<0> base slice", "gc_prep_kernel<0>", 24),
              ("gc_prep_kernel<1> decoder index", "gc_prep_kernel<1>", 8),
              ("gc_dequant_kernel", "gc_dequant", 12)]
    calls = segment(ks)
    out, pos = {}, 1                                       # skip the first lone quantile call
    for label, key, bpe in stages:
        seg = calls[pos:pos + 11]
        assert len(seg) == 11 and all(c[0] == key for c in seg), (label, [c[0] for c in seg])
        t = sum(c[1] for c in seg[1:]) / 10.0 * 1e-9
        out[label] = {"avg_us_per_call": round(t * 1e6, 2), "kernels_per_call": seg[1][2], "algorithmic_bytes_per_element": bpe,
                      "GB_per_s": round(n * bpe / t / 1e9, 1), "frac_of_8TBps": round(n * bpe / t / 8e12, 4)}
        if pmc:
            fs, ws = pmc[0][pos:pos + 11], pmc[1][pos:pos + 11]
            assert all(c[0] == key for c in fs) and all(c[0] == key for c in ws), label
            rd = 2.0 * sum(c[1] for c in fs[1:]) / 10.0 * 1024.0          # FETCH_SIZE in KB, x2 on gfx950 (MI355X_MICROARCH.md)
            wr = sum(c[1] for c in ws[1:]) / 10.0 * 1024.0
            out[label].update({"pmc_read_bytes_per_element": round(rd / n, 2), "pmc_write_bytes_per_element": round(wr / n, 2),
                               "pmc_GB_per_s": round((rd + wr) / t / 1e9, 1), "pmc_frac_of_8TBps": round((rd + wr) / t / 8e12, 4),
                               "pmc_over_algorithmic": round((rd + wr) / (n * bpe), 3)})
        pos += 11
    per_kernel = {}
    for name, ns in ks:
        m = re.search(r"(\w+kernel\w*(<[^>]*>)?)", name)
        a = per_kernel.setdefault(m.group(1) if m else name, [0, 0])
        a[0] += ns; a[1] += 1
    per_kernel = {k: {"calls": v[1], "avg_us": round(v[0] / v[1] * 1e-3, 2)} for k, v in per_kernel.items()}
    from bench import source_hash
    print(json.dumps({"source": f"rocprofv3 --kernel-trace of `python3 tools/stage_bench.py {B}` on one MI355X: one enhancement slice of {B} images of "
                                "1024x1024 (latent 64x64 x 32 channels); kernel durations from the trace, folded by tools/stage_rocprof.py",
                      "elements": n, "stages": out, "kernels": per_kernel, "source_hash": source_hash()}, indent=1))


if __name__ == "__main__":
    main()
