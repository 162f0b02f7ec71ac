#!/usr/bin/env python3
"""Per-kernel sums over the LAST `frames` rollout frames of a rocprofv3 kernel trace (frames are delimited by the first-layer
im2col kernel): launches and us per frame by kernel name, and the frame's span.

    python tools/trace_frame.py t_kernel_trace.csv [frames]
"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 8
starts = [s for s, e, n in rows if "im2col_first" in n]
lo, hi = starts[-nf - 1], starts[-1]
agg = defaultdict(lambda: [0, 0])
busy = 0
for s, e, n in rows:
    if lo <= s < hi:
        n = n.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:70]
        agg[n][0] += e - s
        agg[n][1] += 1
        busy += e - s
print(f"{nf} frames: span {(hi - lo) / nf / 1e3:.1f} us per frame, kernels busy {busy / nf / 1e3:.1f} us per frame")
for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{t / nf / 1e3:9.1f} us  x{c / nf:6.1f}  {n}")
