#!/usr/bin/env python3
"""Per-kernel-family summary of one rocprofv3 PMC pass joined with its kernel trace:

    python tools/collect_pmc.py <counter_collection.csv> <kernel_trace.csv> [out.json] [family-substring]

Prints, per family: launches, mean duration, every collected counter per launch, and the derived figures the guide
defines (MI355X_MICROARCH.md): effective clock = GRBM_GUI_ACTIVE / 8 / wall; matrix-pipe busy = SQ_VALU_MFMA_BUSY_CYCLES
/ (GUI / 8 x 1024 SIMDs); LDS conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE; wave-time shares
SQ_WAIT_ANY, SQ_WAIT_INST_ANY, SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES."""
import collections
import csv
import json
import sys

from collect_traffic import family


def main():
    cc, kt = sys.argv[1:3]
    want = sys.argv[4] if len(sys.argv) > 4 else ""
    dur = {r["Dispatch_Id"]: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9 for r in csv.DictReader(open(kt))}
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    fam_of = {}
    for r in csv.DictReader(open(cc)):
        d = r["Dispatch_Id"]
        fam_of[d] = family(r["Kernel_Name"])
        per[d][r["Counter_Name"]] += float(r["Counter_Value"])
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for d, c in per.items():
        if d not in dur:
            continue
        a = agg[fam_of[d]]
        a["_n"] += 1; a["_t"] += dur[d]
        for k, v in c.items():
            a[k] += v
    res = {}
    for f, a in sorted(agg.items(), key=lambda kv: -kv[1]["_t"]):
        if want and want not in f:
            continue
        n, t = a["_n"], a["_t"]
        e = {"launches": int(n), "avg_ms": 1e3 * t / n}
        for k, v in a.items():
            if not k.startswith("_"):
                e[k + "_per_launch"] = v / n
        gui = a.get("GRBM_GUI_ACTIVE", 0.0)
        if gui:
            e["effective_clock_ghz"] = gui / 8 / t / 1e9
            if "SQ_VALU_MFMA_BUSY_CYCLES" in a:
                e["mfma_busy_frac"] = a["SQ_VALU_MFMA_BUSY_CYCLES"] / (gui / 8 * 1024)
        if a.get("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict_share"] = a.get("SQ_LDS_BANK_CONFLICT", 0.0) / a["SQ_LDS_IDX_ACTIVE"]
        if a.get("SQ_WAVE_CYCLES"):
            for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_BUSY_CYCLES"):
                if k in a:
                    e[k.lower() + "_share"] = a[k] / a["SQ_WAVE_CYCLES"]
        res[f] = e
        print(f, json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in e.items()}))
    if len(sys.argv) > 3 and sys.argv[3] not in ("", "-"):
        json.dump(res, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
