#!/usr/bin/env python3
"""Fold two rocprofv3 PMC passes (one `--pmc FETCH_SIZE`, one `--pmc WRITE_SIZE`, each over
`python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline`) into HBM bytes per launch per kernel family.
FETCH_SIZE / WRITE_SIZE count kilobytes; FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for gfx950 (the counter
reports half of wide coalesced reads).  usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> > profiles/rNN_hbm_traffic.json"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

FAMILIES = [("conv_igemm_uni_kernel", "conv_igemm_uni_kernel"), ("conv_igemm_in_gdn_kernel", "conv_igemm_in_gdn_kernel"), ("conv_igemm_dma_kernel", "conv_igemm_dma_kernel"), ("conv_igemm_kernel", "conv_igemm_kernel"),
            ("gc_prep", "gc_prep/dequant/eb"), ("gc_dequant", "gc_prep/dequant/eb"), ("eb_", "gc_prep/dequant/eb"),
            ("quantile", "quantile"), ("win_attention", "win_attention")]


def fold(d, counter):
    out = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            fam = next((v for k, v in FAMILIES if k in r["Kernel_Name"]), None)
            if fam:
                out[fam][0] += 1
                out[fam][1] += float(r["Counter_Value"])
    for f in glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True):      # rocprofv3's default rocpd (SQLite) output
        import sqlite3
        cur = sqlite3.connect(f).cursor()
        for name, val in cur.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
            fam = next((v for k, v in FAMILIES if k in name), None)
            if fam:
                out[fam][0] += 1
                out[fam][1] += float(val)
    return out


def main():
    fd, wd = sys.argv[1], sys.argv[2]
    rd, wr = fold(fd, "FETCH_SIZE"), fold(wd, "WRITE_SIZE")
    fam = {}
    for k in rd:
        n = rd[k][0]
        r = 2.0 * rd[k][1] * 1024.0 / n
        w = wr[k][1] * 1024.0 / max(1, wr[k][0])
        fam[k] = {"launches_per_2_steps": n, "hbm_read_bytes_per_launch": r, "hbm_write_bytes_per_launch": w, "hbm_bytes_per_launch": r + w}
    from bench import source_hash
    print(json.dumps({"source_hash": source_hash(), "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over `python3 bench.py --steps 1 --warmup 1 "
                                "--no-cpu-baseline`, MI355X; FETCH_SIZE in KB doubled per MI355X_MICROARCH.md (gfx950 reports half of wide "
                                "coalesced reads), WRITE_SIZE in KB as is; folded by tools/pmc_traffic.py", "families": fam}, indent=1))


if __name__ == "__main__":
    main()
