#!/usr/bin/env python3
"""Timeline summary of a rocprofv3 kernel trace CSV: per training step (delimited by adamw_kernel), wall span, union of
busy time, idle gaps, and how much of the span runs >= 2 kernels at once.

    python tools/trace_timeline.py gpurun_out/prof/x_kernel_trace.csv
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
print(f"{len(rows)} dispatches, {len(ends)} optimiser steps")
for i in range(1, len(ends)):
    lo, hi = ends[i - 1], ends[i]
    ks = [(s, e, n, q) for s, e, n, q in rows if s >= lo and e <= hi]
    ev = []
    for s, e, n, q in ks:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    busy = over = 0
    depth = 0
    last = lo
    for t, d in ev:
        if depth >= 1:
            busy += t - last
        if depth >= 2:
            over += t - last
        depth += d
        last = t
    span = hi - lo
    tot = sum(e - s for s, e, n, q in ks)
    perq = defaultdict(int)
    for s, e, n, q in ks:
        perq[q] += e - s
    # gaps on the critical (busy-union) timeline
    gaps = []
    depth = 0
    last = lo
    for t, d in ev:
        if depth == 0 and t > last:
            gaps.append(t - last)
        depth += d
        last = t
    big = sorted(gaps, reverse=True)[:5]
    print(f"step {i}: span {span / 1e6:7.3f} ms  kernels {len(ks)}  sum {tot / 1e6:7.3f}  busy {busy / 1e6:7.3f}  idle {(span - busy) / 1e6:6.3f} "
          f"({len(gaps)} gaps, top {[round(g / 1e3) for g in big]} us)  overlap>=2 {over / 1e6:6.3f}  per-queue "
          + ", ".join(f"q{q}:{v / 1e6:.2f}" for q, v in sorted(perq.items())))
