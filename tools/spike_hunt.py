#!/usr/bin/env python3
"""Which training steps take far longer than the median, and what happened in them: host enqueue time, device time (events),
allocator segments (a new hipMalloc synchronises), Python garbage collections.

    python tools/spike_hunt.py [--steps 80] [--gc off|freeze]
"""
import argparse
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_convlstm_amd as U   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=80)
ap.add_argument("--warmup", type=int, default=8)
ap.add_argument("--gc", default="on")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(1234)
model = U.TemporalUNetDualView(1, 1, base_ch=64, lstm_layers=1, use_skip_lstm=True, use_attention=False).to(dev).train()
opt = U.FusedAdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, max_grad_norm=1.0)
data = U.SyntheticSequences(32, 20, 64, 64, seed=1, kind="uniform", device=dev)
for _ in range(a.warmup):
    U.train_step(model, opt, data.x, data.y, None, False, None)
torch.cuda.synchronize()
if a.gc == "off":
    gc.disable()
elif a.gc == "freeze":
    gc.collect()
    gc.freeze()
gc_log = []
gc.callbacks.append(lambda phase, info: gc_log.append((phase, info["generation"], time.perf_counter())))
marks = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps + 1)]
rows = []
marks[0].record()
for i in range(a.steps):
    st0 = torch.cuda.memory_stats(dev)
    n_gc = len(gc_log)
    t0 = time.perf_counter()
    U.train_step(model, opt, data.x, data.y, None, False, None)
    t1 = time.perf_counter()
    marks[i + 1].record()
    st1 = torch.cuda.memory_stats(dev)
    gcs = [(g, round((gc_log[j + 1][2] - gc_log[j][2]) * 1e3, 1)) for j in range(n_gc, len(gc_log) - 1) for g in [gc_log[j][1]] if gc_log[j][0] == "start"]
    rows.append(((t1 - t0) * 1e3, st1["segment.all.current"] - st0["segment.all.current"], st1["num_device_alloc"] - st0["num_device_alloc"],
                 st1["num_device_free"] - st0["num_device_free"], gcs))
torch.cuda.synchronize()
dts = [marks[i].elapsed_time(marks[i + 1]) for i in range(a.steps)]
med = sorted(dts)[len(dts) // 2]
print(f"median device step {med:.2f} ms, host enqueue median {sorted(r[0] for r in rows)[len(rows) // 2]:.2f} ms, gc={a.gc}")
for i, (dt, r) in enumerate(zip(dts, rows)):
    if dt > 1.15 * med or r[0] > 1.5 * med or r[2] or r[4]:
        print(f"step {i:3d}: device {dt:7.2f} ms  host enqueue {r[0]:7.2f} ms  segments {r[1]:+d}  hipMalloc {r[2]}  hipFree {r[3]}  gc (generation, ms) {r[4]}")
