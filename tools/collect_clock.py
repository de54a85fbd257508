#!/usr/bin/env python3
"""Effective clock and MFMA-pipe utilisation per kernel family from one rocprofv3 PMC pass
(`--pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace`), joined with the kernel trace of the same run.

    python tools/collect_clock.py <counter_collection.csv> <kernel_trace.csv> [out.json]

MI355X_MICROARCH.md: effective clock = GRBM_GUI_ACTIVE / 8 (sum over the 8 XCDs) / kernel wall time (reads high on
dispatches shorter than ~0.3 ms); MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs).
"""
import collections
import csv
import json
import sys

from collect_traffic import family


def main():
    cc, kt = sys.argv[1:3]
    dur = {}
    for r in csv.DictReader(open(kt)):
        dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(cc)):
        f = family(r["Kernel_Name"])
        d = r["Dispatch_Id"]
        acc[(f, d)][r["Counter_Name"]] += float(r["Counter_Value"])
    fam = collections.defaultdict(lambda: dict(n=0, gui=0.0, mfma=0.0, t=0.0))
    for (f, d), c in acc.items():
        if d not in dur:
            continue
        e = fam[f]
        e["n"] += 1; e["gui"] += c.get("GRBM_GUI_ACTIVE", 0.0); e["mfma"] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); e["t"] += dur[d]
    res = {}
    for f, e in sorted(fam.items(), key=lambda kv: -kv[1]["t"]):
        if e["t"] <= 0 or e["gui"] <= 0:
            continue
        res[f] = {"launches": e["n"], "avg_ms": 1e3 * e["t"] / e["n"], "effective_clock_ghz": e["gui"] / 8 / e["t"] / 1e9,
                  "mfma_busy_frac": e["mfma"] / (e["gui"] / 8 * 1024)}
        print(f"{f:<30} n={e['n']:4d} avg {res[f]['avg_ms']:8.3f} ms  clock {res[f]['effective_clock_ghz']:.2f} GHz  MFMA busy {100 * res[f]['mfma_busy_frac']:5.1f} %")
    if len(sys.argv) > 3:
        json.dump(res, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
