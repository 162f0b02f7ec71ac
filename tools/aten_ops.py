#!/usr/bin/env python3
"""Which ATen operators (and from which source lines) still launch kernels inside one training step of the benchmark model:
the hot path is meant to be hand-written kernels only, this lists what is left of PyTorch's own."""
import os
import sys
from collections import Counter

import torch
from torch.profiler import profile, ProfilerActivity

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_convlstm_amd as U   # noqa: E402

torch.manual_seed(0)
B, T = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (8, 20)
model = U.TemporalUNetDualView(1, 1, base_ch=64, use_skip_lstm=True).cuda().train()
opt = U.FusedAdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, max_grad_norm=1.0)
d = U.SyntheticSequences(B, T, 64, 64, seed=1)
for _ in range(2):
    U.train_step(model, opt, d.x, d.y, None, False)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    U.train_step(model, opt, d.x, d.y, None, False)
    torch.cuda.synchronize()
launching = Counter()
for ev in prof.events():
    if not ev.name.startswith("aten::"):
        continue
    if ev.device_time_total <= 0 or any(c.name.startswith("aten::") and c.device_time_total > 0 for c in ev.cpu_children):
        continue          # only leaf ATen ops that own device time
    where = ""
    for fr in ev.stack:
        if "unet-convlstm_amd" in fr or "unet_convlstm_amd" in fr or "bench.py" in fr:
            where = fr.split("unet-convlstm_amd/")[-1]
            break
    launching[(ev.name, str(ev.input_shapes)[:70], where[:80])] += 1
print(f"{sum(launching.values())} ATen ops with device time in one step (B={B}, T={T}):")
for (name, shapes, where), n in launching.most_common(60):
    print(f"  {n:4d}  {name:22s} {shapes:70s} {where}")
