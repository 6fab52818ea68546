#!/usr/bin/env python3
"""Effective shader clock and MFMA-pipe utilisation of the conv launches, from one rocprofv3 pass
`--kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --output-format csv` over bench.py (MI355X_MICROARCH.md, DVFS give-back:
effective clock = GRBM_GUI_ACTIVE / 8 XCDs / kernel wall time, reliable on dispatches of >= 0.3 ms; SQ_VALU_MFMA_BUSY_CYCLES counts
cycles summed over the SIMDs, 64 per v_mfma_f32_32x32x2_f32).  usage: python tools/pmc_clock.py <dir> > profiles/rNN_x_clock_mfma_util.json"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    d = sys.argv[1]
    trace = {}
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            trace[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    ctr = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            ctr.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    rows = []
    for k, (name, ns) in trace.items():
        c = ctr.get(k)
        if not c or "conv_igemm" not in name or "GRBM_GUI_ACTIVE" not in c:
            continue
        rows.append((ns, c["GRBM_GUI_ACTIVE"], c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)))
    big = [r for r in rows if r[0] >= 300000]
    tot_ns, tot_act, tot_mfma = (sum(r[i] for r in rows) for i in range(3))
    clk_big = sum(r[1] for r in big) / 8.0 / max(1, sum(r[0] for r in big))                 # cycles per ns = GHz
    clk_all = tot_act / 8.0 / max(1, tot_ns)
    util = tot_mfma / (1024.0 * tot_act / 8.0) if tot_act else 0.0                           # busy SIMD-cycles / (1024 SIMDs x active cycles)
    from bench import source_hash
    print(json.dumps({"source": "rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES over bench.py, folded by tools/pmc_clock.py",
                      "conv_launches": len(rows), "conv_launches_of_0.3ms_or_more": len(big),
                      "effective_clock_ghz_long_launches": round(clk_big, 3), "effective_clock_ghz_all_launches": round(clk_all, 3),
                      "mfma_busy_simd_cycles": tot_mfma, "gui_active_cycles_per_xcd": tot_act / 8.0,
                      "mfma_pipe_utilisation_while_active": round(util, 4),
                      "f32_mfma_peak_at_this_clock_tflops": round(256 * 256 * clk_big / 1e3, 1), "source_hash": source_hash()}, indent=1))


if __name__ == "__main__":
    main()
