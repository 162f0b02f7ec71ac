#!/usr/bin/env python3
"""How long does the HOST need to enqueue one training step of the benchmark model (no synchronisation inside the loop), against
the GPU's step time?  If the two are close the step is launch-bound wherever the host is slow (step boundaries)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_convlstm_amd as U   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
model = U.TemporalUNetDualView(1, 1, base_ch=64, use_skip_lstm=True).cuda().train()
opt = U.FusedAdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, max_grad_norm=1.0)
d = U.SyntheticSequences(B, 20, 64, 64, seed=1)
for _ in range(3):
    U.train_step(model, opt, d.x, d.y, None, False)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for _ in range(n):
    U.train_step(model, opt, d.x, d.y, None, False)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B}: host enqueue {1e3 * (t1 - t0) / n:.2f} ms per step; GPU finished {1e3 * (t2 - t1):.2f} ms after the last enqueue; "
      f"wall {1e3 * (t2 - t0) / n:.2f} ms per step")
import cProfile
import pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    U.train_step(model, opt, d.x, d.y, None, False)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
