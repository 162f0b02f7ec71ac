#!/usr/bin/env python3
"""Split-K sweep of the per-timestep ConvLSTM GEMMs (developer tool, GPU only): ms / TFLOP/s per ksplit."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_convlstm_amd as U   # noqa: E402,F401
from unet_convlstm_amd import ops  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=20):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


# (name, B*images, H, C_in (= 4*Hd for the recurrent input gradient), N)
SHAPES = [("skip2 dh", 32, 16, 1024, 256), ("skip3 dh", 32, 8, 2048, 512), ("temporal dh", 32, 4, 4096, 1024),
          ("skip3 fwd", 32, 8, 1024, 2048), ("temporal fwd", 32, 4, 2048, 4096)]
for name, B, H, Ci, N in SHAPES:
    x = (torch.randn(B, H, H, Ci, device=DEV) * 0.5).to(torch.bfloat16)
    wp = (torch.randn(N, 9 * Ci, device=DEV) * 0.05).to(torch.bfloat16)
    acc = torch.zeros(B * H * H, N, device=DEV)
    fl = 2.0 * B * H * H * N * 9 * Ci
    line = [f"{name:12s} M={B * H * H:5d} N={N:5d} K={9 * Ci:6d}"]
    for ks in (1, 2, 3, 4, 6, 8, 12, 16, 24):
        ms = timeit(lambda: ops.igemm_atomic([ops.SrcView(x)], wp, (H, H), B, acc, ks, ktap=3, pad=1))
        line.append(f"ks{ks}:{ms * 1e3:6.1f}us/{fl / ms / 1e9:5.0f}")
    print("  ".join(line), flush=True)
