#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, as MI355X_MICROARCH.md section HBM prescribes)
into a per-kernel-family HBM-traffic table: profiles/rNN_hbm_traffic.json, which bench.py reads for `roofline.traffic`.

    python tools/collect_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

Units and corrections (MI355X_MICROARCH.md): FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports exactly
half of the bytes of a wide coalesced streaming read (16 B/lane), so it is doubled; WRITE_SIZE is exact for 16-B
streaming stores.  Our epilogue stores are 4 B/lane (128-B segments), which the guide lists as uncalibrated: the
write figure is therefore reported as-is and flagged.  Values are per launch (mean over the launches of a family).
"""
import collections
import csv
import json
import re
import sys

MATH = {"0": "f32", "1": "bf16", "2": "bf16x3", "3": "f16"}


def family(kernel_name: str) -> str:
    m = re.search(r"tapgemm_kernel<(\d), (\d+), (\d+), (\d), (\d), (\d+)", kernel_name)
    if m:      # bench.py's family name: tapgemm_<math>_<BM>x<BN>[w8]
        return f"tapgemm_{MATH[m.group(1)]}_{m.group(2)}x{m.group(3)}" + ("w8" if int(m.group(4)) * int(m.group(5)) == 8 else "")
    m = re.search(r"lingemm_kernel<(\d+)>", kernel_name)
    if m:      # bench.py's family name: lingemm_bf16_<BM>x128
        return f"lingemm_bf16_{m.group(1)}x128"
    if "gemm256_kernel" in kernel_name:
        return "gemm256_bf16"
    m = re.search(r"gemmcu_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+)(?:, (true|false))?>", kernel_name)
    if m:      # bench.py's family name: gemmcu_{bf16|f16}_<BM>x<BN> (NW, WM, WN, MT, NT, NS, TC: BM = 16 WM MT, BN = 16 WN NT; TC = the fp16 upsamplers)
        return f"gemmcu_{'f16' if m.group(7) == 'true' else 'bf16'}_{16 * int(m.group(2)) * int(m.group(4))}x{16 * int(m.group(3)) * int(m.group(5))}"
    if "attention_bf16" in kernel_name:
        return "attention_bf16"
    # the fused kernels end in <..., ACC, VL>: ACC = the accumulate variant (bench.py's "_acc"), VL = the ragged-batch instantiation
    # (the same family name: bench.py times one or the other in a leg)
    m = re.search(r"reschain_kernel<(true|false), (?:true|false)>", kernel_name)
    if m:      # bench.py's family name: reschain_f16_c32[_acc]
        return "reschain_f16_c32" + ("_acc" if m.group(1) == "true" else "")
    m = re.search(r"respair(?:_wide)?_kernel<(\d+),.*?(true|false), (?:true|false)>", kernel_name)
    if m:      # bench.py's family name: respair_f16_c<C>[_acc]
        return f"respair_f16_c{m.group(1)}" + ("_acc" if m.group(2) == "true" else "")
    m = re.search(r"posconv_kernel<(\d+)>", kernel_name)
    if m:
        return f"posconv_bf16_c{m.group(1)}"
    m = re.search(r"upsample_stream_kernel<(\d+)", kernel_name)
    if m:
        return f"upsample_f16_c{m.group(1)}"
    if "conv_post" in kernel_name:
        return "conv_post"
    return re.sub(r"_kernel.*|\(.*", "", kernel_name).replace("void ", "")


def per_family(path, counter):
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            f = family(r["Kernel_Name"])
            tot[f] += float(r["Counter_Value"])
            n[f] += 1
    return {f: (tot[f] / n[f], n[f]) for f in tot}


def main():
    fetch, write, out = sys.argv[1:4]
    F, W = per_family(fetch, "FETCH_SIZE"), per_family(write, "WRITE_SIZE")
    res = {}
    for f in sorted(set(F) | set(W)):
        fk, nf = F.get(f, (0.0, 0))
        wk, nw = W.get(f, (0.0, 0))
        res[f] = {"launches_sampled": int(max(nf, nw)),
                  "fetch_bytes_per_launch": 2.0 * fk * 1024.0,          # gfx950: FETCH_SIZE counts half of wide coalesced reads
                  "write_bytes_per_launch": wk * 1024.0,
                  "hbm_bytes_per_launch": 2.0 * fk * 1024.0 + wk * 1024.0,
                  "note": "FETCH_SIZE doubled (gfx950 correction); WRITE_SIZE as reported (4-B/lane stores: uncalibrated width)"}
    json.dump(res, open(out, "w"), indent=1)
    for f, v in sorted(res.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_sampled"])[:8]:
        print(f"{f:<28} n={v['launches_sampled']:4d}  fetch {v['fetch_bytes_per_launch'] / 1e6:9.1f} MB  write {v['write_bytes_per_launch'] / 1e6:9.1f} MB per launch")


if __name__ == "__main__":
    main()
