#!/usr/bin/env python3
"""ATen (non-uclstm) kernels of one training step in a rocprofv3 kernel trace: name, launches, total us -- what is still
left to PyTorch on the step's streams.

    python tools/trace_aten.py t_kernel_trace.csv [step_index]
"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
rows.sort()
ends = [e for s, e, n, q in rows if "adamw_kernel" in n]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(ends) - 3
lo, hi = ends[k], ends[k + 1]
agg = defaultdict(lambda: [0, 0])
for s, e, n, q in rows:
    if s >= lo and e <= hi and "at::" in n:
        a = agg[(n[:150], q)]
        a[0] += e - s
        a[1] += 1
for (n, q), (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{t / 1e3:8.1f} us x{c:3d}  q{q}  {n}")
