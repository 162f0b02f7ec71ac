#!/usr/bin/env python3
"""Turn a `rocprofv3 --kernel-trace --stats --output-format csv` kernel_stats.csv into a short markdown table.

    python tools/summarize_rocprof.py gpurun_out/prof/r1_kernel_stats.csv STEPS > profiles/round1_xxx.md
"""
import csv
import sys

path, steps = sys.argv[1], int(sys.argv[2])
rows = list(csv.DictReader(open(path)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"source: `{path}`  ({steps} training steps in the trace)\n")
print(f"total kernel time {tot / 1e6:.2f} ms = {tot / 1e6 / steps:.2f} ms per step\n")
print("| kernel | calls/step | ms/step | avg launch us | % |")
print("|---|---|---|---|---|")
for r in rows:
    pct = float(r["Percentage"])
    if pct < 0.2:
        continue
    name = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    name = name.split("(")[0][:70]
    print(f"| `{name}` | {int(r['Calls']) / steps:.1f} | {float(r['TotalDurationNs']) / 1e6 / steps:.3f} | "
          f"{float(r['AverageNs']) / 1e3:.1f} | {pct:.1f} |")
