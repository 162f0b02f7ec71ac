#!/usr/bin/env python3
"""Dispatch-by-dispatch listing of a rocprofv3 kernel trace around a training-step boundary (the end of adamw_kernel):

    python tools/trace_boundary.py t_kernel_trace.csv [step_index] [us_before] [us_after]

columns: start and end in us relative to the end of the optimiser kernel, hardware queue, kernel.
"""
import csv
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
rows.sort()
ends = [e for s, e, n, q in rows if "adamw_kernel" in n]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(ends) - 3
before = float(sys.argv[3]) if len(sys.argv) > 3 else 800.0
after = float(sys.argv[4]) if len(sys.argv) > 4 else 2500.0
lo = ends[k]
for s, e, n, q in rows:
    if lo - before * 1e3 <= s <= lo + after * 1e3:
        n = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:72]
        print(f"{(s - lo) / 1e3:9.1f} {(e - lo) / 1e3:9.1f}  q{q}  {n}")
