#!/usr/bin/env python3
"""Bitwise run-to-run stability of the ring kernels (igemm_wgrad_c64_kernel / igemm_fwd_c64_kernel) on multi-strip images."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_convlstm_amd as U  # noqa: E402
from unet_convlstm_amd import ops  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n_img = int(sys.argv[2]) if len(sys.argv) > 2 else 48
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 40
dev = "cuda"
torch.manual_seed(3)
x = torch.randn(n_img, W, W, 64, device=dev).relu().to(torch.bfloat16)
dz = (torch.randn(n_img, W, W, 64, device=dev) * 0.1).to(torch.bfloat16)
w = torch.randn(64, 64, 3, 3, device=dev) * 0.05


def wgrad():
    dwp = ops.igemm_wgrad([ops.SrcView(x)], [(dz, 0, 64, 0, 1, 0, 0)], 64, 576, (W, W), n_img, ktap=3, pad=1)
    return dwp.sum(0)          # fixed-order reduction by ATen


def fwd():
    pd = ops.conv_pack_desc(64, 64, [64], [64])
    wp = ops.pack_weights(pd, w)
    out = torch.empty_like(x)
    ops.igemm_store([ops.SrcView(x)], wp, (W, W), n_img, [(out, 0, 64, 0, 1, 0, 0)], ktap=3, pad=1)
    return out


for name, fn in (("wgrad", wgrad), ("fwd", fwd)):
    ops.LAUNCH_LOG = []
    ref = fn()
    log, ops.LAUNCH_LOG = ops.LAUNCH_LOG, None
    bad = 0
    worst = 0.0
    for i in range(iters):
        got = fn()
        if not torch.equal(got, ref):
            bad += 1
            worst = max(worst, float((got.float() - ref.float()).abs().max()))
    torch.cuda.synchronize()
    print(f"stress {name} W={W} n_img={n_img}: kernels {sorted(set(log))}; {bad} of {iters} runs differ from the first (max |diff| {worst:.3e})", flush=True)
