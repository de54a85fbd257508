"""Per-dispatch durations of the fused ResBlock-pair kernels from a rocprofv3 --kernel-trace CSV: launches of one
kernel repeat in the order (k = 3, 7, 11) x (d = 1, 3, 5) within a vocoder pass; prints the mean duration per slot."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(list)
for r in rows:
    n = r["Kernel_Name"]
    if "respair" in n:
        by[n.split("(")[0]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for n, v in by.items():
    v.sort()
    d = [x[1] for x in v]
    slots = collections.defaultdict(list)
    for i, x in enumerate(d):
        slots[i % 9].append(x)
    print(n, len(d), "launches")
    for s in range(9):
        k = (3, 7, 11)[s // 3]; dl = (1, 3, 5)[s % 3]
        xs = slots[s][1:] or slots[s]
        print(f"   k={k:2d} d={dl}: {sum(xs) / len(xs) / 1e3:8.1f} us  (n={len(xs)})")
