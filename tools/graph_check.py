#!/usr/bin/env python3
"""GraphedTrainStep vs eager from the same seed: loss and parameter difference after 2 eager + 1 replayed step.
    [SINGLE_STREAM=1] python tools/graph_check.py BASE_CH B T"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_convlstm_amd as U
from unet_convlstm_amd import ops
base, B, T = (int(v) for v in sys.argv[1:4])
if os.environ.get("SINGLE_STREAM"):
    ops.ASYNC_WGRAD = False; ops.BN_RUNNING_ON_SIDE = False; ops.PARAM_GRADS_ON_SIDE = False
if os.environ.get("NO_PREPACK"):
    ops.PREPACK = False
def make(cap):
    torch.manual_seed(5)
    m = U.TemporalUNetDualView(1, 1, base_ch=base, use_skip_lstm=True).cuda().train()
    return m, U.FusedAdamW(m.parameters(), lr=1e-3, weight_decay=1e-4, max_grad_norm=1.0, capturable=cap)
d = U.SyntheticSequences(B, T, 64, 64, seed=6, kind="uniform")
m1, o1 = make(False); m2, o2 = make(True); m3, o3 = make(True)
for _ in range(3):
    l1, _ = U.train_step(m1, o1, d.x, d.y, d.mask, True)
for _ in range(3):
    l3, _ = U.train_step(m3, o3, d.x, d.y, d.mask, True)          # capturable optimiser, eager
g = U.GraphedTrainStep(m2, o2, d.x, d.y, d.mask, True, warmup=2)
l2, _ = g(d.x, d.y, d.mask)
torch.cuda.synchronize()
r = lambda a, b: float((a - b).norm() / b.norm())
print(f"graph_check base={base} B={B} T={T} single={bool(os.environ.get('SINGLE_STREAM'))} noprepack={bool(os.environ.get('NO_PREPACK'))}: "
      f"loss eager {float(l1):.7f} eager-capturable {float(l3):.7f} graph {float(l2):.7f}; params: capturable-eager vs eager {r(o3.flat.flat_p, o1.flat.flat_p):.2e}, "
      f"graph vs eager {r(o2.flat.flat_p, o1.flat.flat_p):.2e}", flush=True)
