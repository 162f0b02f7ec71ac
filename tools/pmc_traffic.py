#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; MI355X_MICROARCH.md section HBM).

    python tools/pmc_traffic.py gpurun_out/pmc_f/f_counter_collection.csv gpurun_out/pmc_w/w_counter_collection.csv [out.json]

bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KiB and on gfx950 FETCH_SIZE reports half the bytes
of a wide (16 B/lane) coalesced read stream, which is how every kernel here reads.  Kernels are grouped by family.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def family(name: str) -> str:
    m = re.search(r"igemm_fwd_kernel<(\d), (\d), (\d)", name)
    if m:          # same keys as bench.py's roofline leg: epilogue family[tile shape]
        epi = {"0": "igemm_fwd_store", "1": "igemm_fwd_lstm", "2": "igemm_fwd_atomic"}[m.group(1)]
        return epi + "[" + {"0": "pertap128x128", "1": "pertap64x256", "2": "patch128x256"}[m.group(2)] + "]"
    if "igemm_fwd_group_kernel" in name:   # several independent patch-shape GEMMs in one launch (forward recurrence)
        return "igemm_fwd_group[patch128x256]"
    if "igemm_fwd_c64" in name:        # the persistent 64-channel kernel serves UCLSTM_EPI_STORE launches
        return "igemm_fwd_store[ring64]"
    if "igemm_wgrad_c64" in name:
        return "igemm_wgrad[ring64]"
    if "igemm_wgrad_p3" in name:
        return "igemm_wgrad[p3_256x256]"
    m = re.search(r"igemm_wgrad_p2_kernel<(\d)", name)
    if m:
        return "igemm_wgrad[p2_64x256]" if m.group(1) == "1" else "igemm_wgrad[p2_128x128]"
    if "igemm_wgrad" in name:
        return "igemm_wgrad[generic]"
    m = re.search(r"(\w+_kernel)", name)
    return m.group(1) if m else name[:40]


def load(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] != counter:
                continue
            a = acc[family(r["Kernel_Name"])]
            a[0] += float(r["Counter_Value"])
            a[1] += 1
    return acc


fe, wr = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fe) | set(wr), key=lambda k: -(2 * fe[k][0] + wr[k][0])):
    n = max(fe[k][1], wr[k][1], 1)
    rd, wt = 2 * fe[k][0] * 1024 / n, wr[k][0] * 1024 / n
    out[k] = {"launches": n, "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wt), "hbm_bytes_per_launch": round(rd + wt)}
tot = sum(v["hbm_bytes_per_launch"] * v["launches"] for v in out.values())
print(f"total HBM bytes in the trace: {tot / 1e9:.2f} GB")
for k, v in list(out.items())[:25]:
    print(f"{k:32s} x{v['launches']:5d}  read {v['read_bytes_per_launch'] / 1e6:9.2f} MB  write {v['write_bytes_per_launch'] / 1e6:9.2f} MB per launch")
if len(sys.argv) > 3:
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024",
               "kernels": out}, open(sys.argv[3], "w"), indent=1)
