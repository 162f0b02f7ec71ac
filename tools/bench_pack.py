#!/usr/bin/env python3
"""Pack / unpack kernel throughput on the benchmark model's panels (GB/s of f32 read + bf16 written)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_convlstm_amd as U   # noqa: E402
from unet_convlstm_amd import ops   # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


cases = []
for name, Hd in (("temporal", 1024), ("skip3", 512), ("skip2", 256)):
    w = torch.randn(4 * Hd, 2 * Hd, 3, 3, device="cuda")
    cases += [(f"{name} lstm fwd", ops.lstm_pack_desc(Hd, Hd), w, 0), (f"{name} lstm dgrad h", ops.lstm_dgrad_pack_desc(Hd, Hd, Hd), w, Hd * 9),
              (f"{name} lstm dgrad x", ops.lstm_dgrad_pack_desc(Hd, Hd, Hd), w, 0)]
for co, ci in ((1024, 1024), (512, 1024), (512, 512), (256, 256), (128, 128), (64, 64)):
    w = torch.randn(co, ci, 3, 3, device="cuda")
    cases += [(f"conv {ci}->{co} fwd", ops.conv_pack_desc(co, ci, [ci], [ci]), w, 0), (f"conv {ci}->{co} dgrad", ops.conv_dgrad_pack_desc(co, ci, ci), w, 0)]
tot_t = tot_b = 0.0
for name, d, w, off in cases:
    t = timeit(lambda: ops.pack_weights(d, w, off))
    nbytes = d.N * d.Ktot * 2 + d.N * d.Ktot * 4 * 1.0
    tot_t += t
    tot_b += nbytes
    print(f"{name:28s} panel {d.N:5d} x {d.Ktot:6d}  {t * 1e6:8.1f} us  {nbytes / t / 1e9:8.1f} GB/s")
print(f"sum {tot_t * 1e3:.3f} ms, {tot_b / tot_t / 1e9:.1f} GB/s")
