"""Idle time between kernels: reads a rocprofv3 --kernel-trace CSV of `bench.py` and reports, for the busiest stretch of
back-to-back launches (the timed steps), the sum of kernel durations, the wall time they span and the gaps between one kernel's
end and the next one's start.  usage: python tools/trace_gaps.py <kernel_trace.csv> [launches_per_step]"""
import csv, sys
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# stretches: consecutive kernels less than 200 us apart
stretches, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if b[0] - a[1] > 200_000:
        stretches.append(cur); cur = []
    cur.append(b)
stretches.append(cur)
best = max(stretches, key=len)
busy = sum(e - s for s, e, _ in best)
span = best[-1][1] - best[0][0]
gaps = [max(0, b[0] - a[1]) for a, b in zip(best, best[1:])]
overl = sum(1 for a, b in zip(best, best[1:]) if b[0] < a[1])
print(f"{len(best)} kernels back to back: span {span / 1e6:.3f} ms, busy {busy / 1e6:.3f} ms, gaps {sum(gaps) / 1e6:.3f} ms "
      f"({sum(gaps) / len(gaps) / 1e3:.2f} us per launch on average, median {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us, max {max(gaps) / 1e3:.1f} us; {overl} overlapping pairs)")
by = {}
for (a, b) in zip(best, best[1:]):
    k = a[2][:60]
    g = max(0, b[0] - a[1])
    by.setdefault(k, [0, 0]); by[k][0] += g; by[k][1] += 1
for k, (g, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:12]:
    print(f"  after {k:<60} {n:5d} x {g / n / 1e3:6.2f} us")
