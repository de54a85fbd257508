"""Durations of every launch of the kernels matching a substring, in launch order, for the LAST complete step of a
rocprofv3 --kernel-trace CSV (usage: trace_family.py <kernel_trace.csv> <substring> <launches per step>)."""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[3])
last = rows[-n:]
print(len(rows), "launches; last", n, ":")
print(" ".join(f"{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:.1f}" for r in last))
g = [(r.get("Grid_Size_X", r.get("Grid_Size", "0")), r.get("Workgroup_Size_X", r.get("Workgroup_Size", "1"))) for r in last]
print("grids:", " ".join(f"{int(a) // int(b)}" for a, b in g))
