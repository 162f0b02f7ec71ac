#!/usr/bin/env python3
"""The reference's overfit check (train/overfit_check.py:36-122, custom-UNet branch) run UNCHANGED on the MI355X path:
16 sequences, TemporalUNetDualView(base_ch=64, use_skip_lstm=True), the reference's own loop -- zero_grad, model(x),
torch.stack(output, dim=1), masked MSE written in plain torch on the f32 outputs, loss.backward(), torch.optim.AdamW(lr 1e-3,
wd 1e-4).step() -- until the reference's pass criterion: masked MSE < 5e-4 within 3001 iterations (:91, :116).

The .npz the reference trains on is not available offline; 16 seeded Moving-MNIST-shaped blob sequences stand in.
    python tools/overfit_check.py [--fused] [--dtype f16] [--size 64] [--seq 8]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_convlstm_amd as U   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=64)
ap.add_argument("--seq", type=int, default=8)
ap.add_argument("--iters", type=int, default=3001)
ap.add_argument("--fused", action="store_true", help="FusedAdamW + train-step helper instead of the reference's torch.optim.AdamW loop")
ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
a = ap.parse_args()
device = torch.device("cuda")
if a.dtype == "f16":
    U.set_compute_dtype(torch.float16)
torch.manual_seed(0)
num_samples = 16                                                     # overfit_check.py:42
d = U.SyntheticSequences(num_samples, a.seq, a.size, a.size, seed=11, kind="blobs", device=device)
x, y, mask = d.x, d.y, d.mask
print(f"Batch Shapes -> X: {tuple(x.shape)}, Y: {tuple(y.shape)}")
model = U.TemporalUNetDualView(in_channels_per_sat=1, out_channels=1, base_ch=64, lstm_layers=1, use_skip_lstm=True,
                               use_attention=False).to(device)     # overfit_check.py:74-81
if a.fused or a.dtype == "f16":
    optimizer = U.FusedAdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, loss_scale=2.0 ** 14 if a.dtype == "f16" else None)
else:
    optimizer = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4)      # overfit_check.py:83
t0 = time.time()
ok = False
for i in range(a.iters):                                            # overfit_check.py:91
    optimizer.zero_grad()
    output, _ = model(x)
    y_pred = torch.stack(output, dim=1) if isinstance(output, list) else output
    diff = (y_pred - y) ** 2
    loss = (diff * mask).sum() / (mask.sum() + 1e-6)                 # overfit_check.py:106-107
    (optimizer.scale_loss(loss) if hasattr(optimizer, "scale_loss") else loss).backward()
    optimizer.step()
    if i % 100 == 0:
        lv = loss.item()
        print(f"Iter {i:04d} | Loss: {lv:.6f} | {time.time() - t0:.1f} s", flush=True)
        if lv < 0.0005:                                              # overfit_check.py:116
            print(f"\n[SUCCESS] Loss is near zero after {i} iterations ({time.time() - t0:.1f} s)")
            ok = True
            break
if not ok:
    raise SystemExit("[WARNING] Did not reach the reference's criterion (masked MSE < 5e-4 within 3001 iterations)")
