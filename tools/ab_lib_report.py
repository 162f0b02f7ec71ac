#!/usr/bin/env python3
"""Average launch time of selected kernels from the runs of tools/ab_lib_profile.sh:  python tools/ab_lib_report.py TAG kernel [kernel ...]"""
import csv
import glob
import re
import sys

tag, names = sys.argv[1], sys.argv[2:]
for d in sorted(glob.glob(f"gpurun_out/r3ab_{tag}_*_?")):
    print("==", d.split("r3ab_")[1])
    tot = 0.0
    for r in csv.DictReader(open(d + "/s_kernel_stats.csv")):
        n = re.sub(r"\(.*", "", r["Name"].replace("(anonymous namespace)::", "").replace("void ", ""))
        tot += float(r["TotalDurationNs"])
        if n in names:
            print(f"  {n:28s} calls {r['Calls']:>4s}  avg {float(r['AverageNs']):9.1f} ns")
    print(f"  all kernels: {tot / 21 / 1e6:.3f} ms per step")
