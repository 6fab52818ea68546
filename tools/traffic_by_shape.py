#!/usr/bin/env python3
"""Per-shape HBM-side traffic of the conv family (VERDICT r02 "Next round" 5): joins the per-dispatch `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE`
counters of two serial bench passes with the per-launch shape table (PC_PROFILE_CSV) of the same kind of step.

The last step a `bench.py --lean --overlap 0` process runs is the roofline-profile step: one stream, one lane, 542 conv launches (543 until round 3) in
launch order -- the order of the CSV rows.  So the LAST len(csv) conv_igemm* dispatches of each PMC pass are that step's launches, row
by row.  FETCH_SIZE (KB) is doubled as MI355X_MICROARCH.md prescribes for gfx950; both counters sit on the L2's memory side, so
Infinity-Cache hits are included: these are L2 misses, an upper bound of the HBM bytes.

Times come from a launch table written WITHOUT a profiler attached (counter collection serialises the dispatches and stretches them): the
optional fourth argument, same launch order.

usage: python tools/traffic_by_shape.py <fetch_dir> <write_dir> <launch csv of the PMC pass> [<launch csv of an unprofiled run>] > profiles/rNN_traffic_by_shape.json"""
import collections
import csv
import glob
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_dispatch(d, counter):
    db = glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select dispatch_id, kernel_name, value from counters_collection where counter_name = ? order by dispatch_id", (counter,)).fetchall()
    return [(n, float(v)) for _, n, v in rows if "conv_igemm" in n]


def main():
    fd, wd, cf = sys.argv[1:4]
    launches = list(csv.DictReader(open(cf)))
    n = len(launches)
    if len(sys.argv) > 4:                                      # durations of an unprofiled run of the same step
        timed = list(csv.DictReader(open(sys.argv[4])))
        assert len(timed) == n and all((a["M"], a["N"], a["K"], a["epi"]) == (b["M"], b["N"], b["K"], b["epi"]) for a, b in zip(launches, timed))
        for a, b in zip(launches, timed):
            a["us"] = b["us"]
    f, w = per_dispatch(fd, "FETCH_SIZE")[-n:], per_dispatch(wd, "WRITE_SIZE")[-n:]
    assert len(f) == n and len(w) == n, (len(f), len(w), n)
    agg = collections.OrderedDict()
    for row, (kn, fk), (kn2, wk) in zip(launches, f, w):
        key = (int(row["M"]), int(row["N"]), int(row["K"]), int(row["nphase"]), int(row["epi"]), float(row["gflop"]))
        a = agg.setdefault(key, dict(launches=0, us=0.0, alg_mb=0.0, fetch_mb=0.0, write_mb=0.0, kernel=kn.split("(")[0][-60:]))
        a["launches"] += 1
        a["us"] += float(row["us"])
        a["alg_mb"] += float(row["alg_mbytes"])
        a["fetch_mb"] += 2.0 * fk * 1024.0 / 1e6
        a["write_mb"] += wk * 1024.0 / 1e6
    shapes = []
    for (M, N, K, nph, epi, gf), a in sorted(agg.items(), key=lambda kv: -kv[1]["us"]):
        L = a["launches"]
        tr = (a["fetch_mb"] + a["write_mb"]) / L
        shapes.append({"M": M, "N": N, "K": K, "nphase": nph, "epi": epi, "gflop_per_launch": gf, "launches": L, "ms_per_step": round(a["us"] / 1e3, 3),
                       "tflops": round(gf * L / a["us"] * 1e3, 1), "algorithmic_mb_per_launch": round(a["alg_mb"] / L, 2),
                       "fetch_mb_per_launch": round(a["fetch_mb"] / L, 2), "write_mb_per_launch": round(a["write_mb"] / L, 2),
                       "traffic_over_algorithmic": round(tr / (a["alg_mb"] / L), 2)})
    tot_alg = sum(a["alg_mb"] for a in agg.values())
    tot_tr = sum(a["fetch_mb"] + a["write_mb"] for a in agg.values())
    heads = [s for s in shapes if s["M"] == 8192 and s["N"] == 224]
    h_alg = sum(s["algorithmic_mb_per_launch"] * s["launches"] for s in heads)
    h_tr = sum((s["fetch_mb_per_launch"] + s["write_mb_per_launch"]) * s["launches"] for s in heads)
    from bench import source_hash
    print(json.dumps({"source_hash": source_hash(),
                      "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `bench.py --steps 1 --warmup 1 --lean --overlap 0` (serial form), "
                                "last 542 conv dispatches = the roofline-profile step, joined by launch order with PC_PROFILE_CSV; FETCH_SIZE x2 (gfx950); "
                                "L2 misses incl. Infinity-Cache hits",
                      "launches": n, "traffic_mb_per_launch": round(tot_tr / n, 2), "algorithmic_mb_per_launch": round(tot_alg / n, 2),
                      "traffic_over_algorithmic": round(tot_tr / tot_alg, 3),
                      "slice_chain_heads_traffic_over_algorithmic": round(h_tr / h_alg, 3) if h_alg else None,
                      "shapes": shapes}, indent=1))


if __name__ == "__main__":
    main()
