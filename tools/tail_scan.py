#!/usr/bin/env python3
"""Scan a rocprofv3 kernel trace for launches whose grid needs a SMALL extra round beyond what the chip holds at once.

    python tools/tail_scan.py gpurun_out/xxx/a_kernel_trace.csv [steps]

Residency per CU from the launch's own resources (512 VGPRs per SIMD lane-slice, 4 SIMDs, 160 KiB of LDS, at most 8 waves per SIMD as
the arch+accum VGPR granule allows); rounds = blocks / (256 CUs x blocks per CU).  Flags launches with 1 < rounds and a last round
filled to less than 35 %, or fewer blocks than CUs, sorted by the time they take per step.  An estimate (it ignores that blocks of one
launch finish at different times), meant to point at grids worth an in-step A/B.
"""
import collections
import csv
import math
import re
import sys

path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 21
CUS = 256
agg = collections.OrderedDict()
for r in csv.DictReader(open(path)):
    name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""))[:48]
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    blocks = (int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])) // max(wg, 1)
    vg = int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"])
    lds = int(r["LDS_Block_Size"])
    waves_blk = max(1, wg // 64)
    waves_simd = min(8, 512 // max(8, ((vg + 7) // 8) * 8))            # waves per SIMD the register file allows
    per_cu = max(1, (waves_simd * 4) // waves_blk)
    if lds > 0:
        per_cu = max(1, min(per_cu, (160 * 1024) // lds))
    key = (name, blocks, vg, lds, wg)
    a = agg.setdefault(key, [0, 0.0, per_cu])
    a[0] += 1
    a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
out = []
for (name, blocks, vg, lds, wg), (n, us, per_cu) in agg.items():
    rounds = blocks / (CUS * per_cu)
    frac = rounds - math.floor(rounds)
    flag = ""
    if blocks < CUS:
        flag = f"only {blocks} blocks"
    elif rounds > 1 and 0 < frac < 0.35:
        flag = f"last round {frac:.0%} full"
    if flag:
        out.append((us / steps, name, blocks, wg, vg, lds, per_cu, rounds, n / steps, us / n, flag))
print(f"{'kernel':48s} {'blocks':>7s} {'wg':>4s} {'vgpr':>4s} {'lds':>6s} {'/CU':>3s} {'rounds':>7s} {'n/step':>6s} {'avg us':>8s} {'us/step':>8s}  note")
for t, name, blocks, wg, vg, lds, per_cu, rounds, n, avg, flag in sorted(out, reverse=True)[:40]:
    print(f"{name:48s} {blocks:7d} {wg:4d} {vg:4d} {lds:6d} {per_cu:3d} {rounds:7.2f} {n:6.1f} {avg:8.1f} {t:8.1f}  {flag}")
