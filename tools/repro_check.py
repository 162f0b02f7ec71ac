#!/usr/bin/env python3
"""Run-to-run reproducibility of one training step: N runs from the same state, per-tensor worst rel-L2 against run 0.

    python tools/repro_check.py --size 256 --seq 12 --batch 4 [--dtype f16] [--runs 6]
Environment switches (UCLSTM_WGRAD_RING=0, ...) are read by the package at import, so set them on the command line.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_convlstm_amd as U  # noqa: E402
from unet_convlstm_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=256)
ap.add_argument("--seq", type=int, default=12)
ap.add_argument("--batch", type=int, default=4)
ap.add_argument("--runs", type=int, default=6)
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--sync-wgrad", action="store_true")
a = ap.parse_args()
if a.sync_wgrad:
    ops.ASYNC_WGRAD = False
dt = torch.float16 if a.dtype == "f16" else torch.bfloat16
torch.manual_seed(31)
with ops.compute_dtype(dt):
    model = U.TemporalUNetDualView(1, 1, base_ch=64, use_skip_lstm=True).cuda().train()
    data = U.SyntheticSequences(a.batch, a.seq, a.size, a.size, seed=32, kind="uniform")
    opt = U.FusedAdamW(model.parameters(), lr=0.0, weight_decay=0.0, max_grad_norm=None, loss_scale=2.0 ** 14 if dt == torch.float16 else None)
    grads = []
    for i in range(a.runs):
        loss, _ = U.train_step(model, opt, data.x, data.y, data.mask, True, clip_norm=None)
        grads.append(opt.flat.flat_g.detach().clone())
    torch.cuda.synchronize()
worst = {}
for g in grads[1:]:
    for (k, p), o in zip(model.named_parameters(), opt.flat.offsets):
        a0, b0 = grads[0][o:o + p.numel()].double(), g[o:o + p.numel()].double()
        e = float((a0 - b0).norm() / (a0.norm() + 1e-30))
        worst[k] = max(worst.get(k, 0.0), e)
top = sorted(worst.items(), key=lambda kv: -kv[1])[:6]
env = {k: v for k, v in os.environ.items() if k.startswith("UCLSTM_")}
print(f"repro {a.size}x{a.size} T={a.seq} B={a.batch} {a.dtype} env={env} sync_wgrad={a.sync_wgrad}: " + ", ".join(f"{k} {v:.1e}" for k, v in top), flush=True)
