#!/usr/bin/env python3
"""Per-dispatch rocprofv3 counters of the kernels whose name contains <substr>, grouped by (kernel, grid size) -- the launches of
one instantiation on different GEMM shapes have different grids:  python tools/pmc_by_grid.py <counter_collection.csv> <kernel_trace.csv> <substr>"""
import collections
import csv
import sys

cc, kt, want = sys.argv[1:4]
dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3 for r in csv.DictReader(open(kt))}
per = collections.defaultdict(dict)
key = {}
for r in csv.DictReader(open(cc)):
    if want not in r["Kernel_Name"]:
        continue
    d = r["Dispatch_Id"]
    key[d] = (r["Kernel_Name"].split("(")[0][-40:], r.get("Grid_Size", "?"))
    per[d][r["Counter_Name"]] = per[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for d, c in per.items():
    a = agg[key[d]]
    a["_n"] += 1
    a["_us"] += dur.get(d, 0.0)
    for k, v in c.items():
        a[k] += v
for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["_us"]):
    n = a["_n"]
    line = f"{k[0]} grid {k[1]}: {int(n)} launches, {a['_us'] / n:.1f} us"
    wc = a.get("SQ_WAVE_CYCLES", 0.0)
    for c in sorted(a):
        if c.startswith("_"):
            continue
        line += f"; {c} {a[c] / n:.4g}" + (f" ({a[c] / wc:.3f} of wave cycles)" if wc and c.startswith("SQ_") and c != "SQ_WAVE_CYCLES" else "")
    gui = a.get("GRBM_GUI_ACTIVE", 0.0)
    if gui and a["_us"]:
        line += f"; clock {gui / 8 / (a['_us'] * 1e-6) / 1e9:.2f} GHz"
        if "SQ_VALU_MFMA_BUSY_CYCLES" in a:
            line += f"; mfma busy {a['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui / 8 * 1024):.3f}"
    try:
        print(line)
    except BrokenPipeError:
        break
