#!/usr/bin/env python3
"""Per-queue view of a rocprofv3 kernel trace: for the last full training step (delimited by adamw_kernel) print, per
hardware queue, the busy time, the idle gaps and the per-kernel sums -- the main stream's queue is the critical path, the
others (weight gradients, look-ahead packing, collectives) only matter where they stretch it.

    python tools/trace_queues.py gpurun_out/prof/x_kernel_trace.csv [step_index]
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
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(ends) - 2
lo, hi = ends[k], ends[k + 1]
ks = [(s, e, n, q) for s, e, n, q in rows if s >= lo and e <= hi]
print(f"step {k}: span {(hi - lo) / 1e6:.3f} ms, {len(ks)} dispatches")


def short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:60]


byq = defaultdict(list)
for s, e, n, q in ks:
    byq[q].append((s, e, n))
for q, lst in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e, _ in kv[1])):
    busy = sum(e - s for s, e, _ in lst)
    first, last = lst[0][0], lst[-1][1]
    gaps = [lst[i + 1][0] - lst[i][1] for i in range(len(lst) - 1)]
    pos = [g for g in gaps if g > 0]
    print(f"\nqueue {q}: {len(lst)} kernels, busy {busy / 1e6:.3f} ms, active window {(last - first) / 1e6:.3f} ms, "
          f"gaps inside the window {sum(pos) / 1e6:.3f} ms (median {sorted(pos)[len(pos) // 2] / 1e3 if pos else 0:.1f} us, "
          f"{sum(1 for g in pos if g > 20000)} above 20 us)")
    agg = defaultdict(lambda: [0, 0])
    for s, e, n in lst:
        a = agg[short(n)]
        a[0] += e - s
        a[1] += 1
    for n, (t, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:28]:
        print(f"    {t / 1e6:7.3f} ms  x{c:4d}  {n}")
    big = sorted(((lst[i + 1][0] - lst[i][1], i) for i in range(len(lst) - 1)), reverse=True)[:6]
    for g, i in big:
        if g > 20000:
            print(f"    gap {g / 1e3:7.1f} us  at +{(lst[i][1] - lo) / 1e6:6.3f} ms  after {short(lst[i][2])[:40]}  before {short(lst[i + 1][2])[:40]}")
