#!/usr/bin/env python3
"""Device time of the phases of a training step (forward+loss / backward / clip+AdamW), HIP events on the main stream.

    UCLSTM_GROUP_LSTM=0 python tools/phase_times.py [--batch 32] [--steps 20]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_convlstm_amd as U  # noqa: E402
from unet_convlstm_amd import ops  # noqa: E402
from unet_convlstm_amd.engine import _stack  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--seq", type=int, default=20)
ap.add_argument("--size", type=int, default=64)
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
torch.manual_seed(1234)
model = U.TemporalUNetDualView(1, 1, base_ch=64, use_skip_lstm=True).cuda().train()
opt = U.FusedAdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, max_grad_norm=1.0)
d = U.SyntheticSequences(a.batch, a.seq, a.size, a.size, seed=1, kind="uniform")
tot = [0.0, 0.0, 0.0]
ev = lambda: torch.cuda.Event(enable_timing=True)
for it in range(a.steps + 5):
    if it == 5:
        U.quiesce_host_gc()
        torch.cuda.synchronize()
    e = [ev() for _ in range(4)]
    opt.zero_grad()
    ops.prepack_begin()
    e[0].record()
    out, _ = model(d.x)
    loss = U.compute_loss(_stack(out), d.y, None, False)
    e[1].record()
    loss.backward()
    e[2].record()
    ops.prepack_end()
    opt.step()
    e[3].record()
    if it >= 5:
        torch.cuda.synchronize()
        for k in range(3):
            tot[k] += e[k].elapsed_time(e[k + 1])
n = a.steps
env = {k: v for k, v in os.environ.items() if k.startswith("UCLSTM_")}
print(f"phases B={a.batch} {env}: forward {tot[0] / n:.3f} ms  backward {tot[1] / n:.3f} ms  optimiser {tot[2] / n:.3f} ms  sum {sum(tot) / n:.3f} ms", flush=True)
