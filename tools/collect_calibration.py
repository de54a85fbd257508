#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE per known byte for each access width (tools/ubench/traffic_calib.hip):

    python tools/collect_calibration.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

Every kernel of the micro-benchmark reads 512 MiB and writes 512 MiB; the counters are in KiB."""
import collections, csv, json, re, sys

BYTES = 512 << 20


def per_kernel(path, counter):
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        m = re.search(r"copy_kernel<(.*?)>", k)
        name = {"unsigned int __vector(4)": "copy16", "unsigned int __vector(2)": "copy8", "unsigned int": "copy4"}.get(m.group(1), m.group(1)) if m else ("rows8" if "rows8" in k else None)
        if name:
            tot[name] += float(r["Counter_Value"]); n[name] += 1
    return {k: tot[k] / n[k] for k in tot}


def main():
    f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    res = {}
    for k in sorted(set(f) | set(w)):
        res[k] = {"known_bytes_read": BYTES, "known_bytes_written": BYTES,
                  "FETCH_SIZE_bytes": f.get(k, 0.0) * 1024, "WRITE_SIZE_bytes": w.get(k, 0.0) * 1024,
                  "fetch_reported_over_actual": f.get(k, 0.0) * 1024 / BYTES, "write_reported_over_actual": w.get(k, 0.0) * 1024 / BYTES}
        print(f"{k:8s} FETCH_SIZE reports {res[k]['fetch_reported_over_actual']:.3f} x the bytes read, WRITE_SIZE {res[k]['write_reported_over_actual']:.3f} x the bytes written")
    json.dump(res, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
