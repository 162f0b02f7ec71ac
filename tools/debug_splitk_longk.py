#!/usr/bin/env python3
"""Diagnose test_split_k_store_convolution_equals_the_single_pass[520-2048-1024-4-4-bf16]: which elements differ by more than one bf16
ulp of their own magnitude between the split-K and the one-pass store, and which path is closer to an f64 reference there?"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_convlstm_amd as U   # noqa: E402,F401
from unet_convlstm_amd import ops  # noqa: E402

DEV = "cuda"
N, Ci, Co, H, W = 520, 2048, 1024, 4, 4
dt = torch.bfloat16
torch.manual_seed(31)
x = torch.randn(N, Ci, H, W).to(dt).float()
w = (torch.randn(Co, Ci, 3, 3) * 0.05).to(dt).float()
b, sc, sh = torch.randn(Co), torch.rand(Co) + 0.5, torch.randn(Co) * 0.3
xa = x.permute(0, 2, 3, 1).to(dt).to(DEV).contiguous()
pd = ops.conv_pack_desc(Co, Ci, [Ci], [Ci])
wp = ops.pack_weights(pd, w.to(DEV), 0, dt)
bp, scp, shp = b.to(DEV), sc.to(DEV), sh.to(DEV)
outs = {}
for split in (True, False):
    ops.SPLITK_STORE = split
    out = torch.empty((N, H, W, Co), dtype=dt, device=DEV)
    ops.igemm_store([ops.SrcView(xa)], wp, (H, W), N, [(out, 0, Co, 0, 1, 0, 0)], ktap=3, pad=1, bias=bp, col_scale=scp, col_shift=shp, relu=True)
    outs[split] = out.float().cpu()
ops.SPLITK_STORE = True
a, c = outs[True], outs[False]
# f64 reference on a subset of images (the full one is 314 GFLOP in f64 on the host)
sub = slice(0, 40)
pre64 = F.conv2d(x[sub].double(), w.double(), b.double(), padding=1)
ref64 = torch.relu(pre64 * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)).permute(0, 2, 3, 1)
mag = (F.conv2d(x[sub].abs(), w.abs(), None, padding=1) * sc.view(1, -1, 1, 1)).permute(0, 2, 3, 1)      # sum of |products| x scale
ulp = 2.0 ** -7
diff = (a - c).abs()
bad = diff > ulp * c.abs().clamp_min(2.0 ** -10) * 1.01
print(f"elements {a.numel()}, differing {int((diff > 0).sum())} ({float((diff > 0).float().mean()):.4%}), beyond one ulp of |c|: {int(bad.sum())}")
print(f"typical |c| (median of non-zero) {float(c[c > 0].median()):.3f}, max {float(c.max()):.2f}; median sum|products|*scale {float(mag.median()):.1f}")
bi = bad.nonzero()
for idx in bi[:12]:
    i = tuple(int(v) for v in idx)
    print(f"  {i}: split {float(a[i]):.6g}  one-pass {float(c[i]):.6g}  diff {float(diff[i]):.3g} = {float(diff[i]) / max(float(c[i]), 2**-10) / ulp:.1f} ulp of |c|")
print(f"violators: max |c| {float(c[bad].max()) if bad.any() else 0:.4g}, max diff {float(diff[bad].max()) if bad.any() else 0:.3g}; "
      f"largest diff overall {float(diff.max()):.4g} at |c| = {float(c.flatten()[diff.flatten().argmax()]):.4g}")
as_, cs_, bs_ = a[sub].double(), c[sub].double(), bad[sub]
print(f"vs f64 on images 0..39: rel-L2 split {float((as_ - ref64).norm() / ref64.norm()):.3e}  one-pass {float((cs_ - ref64).norm() / ref64.norm()):.3e}")
if bs_.any():
    ea, ec = (as_ - ref64).abs()[bs_], (cs_ - ref64).abs()[bs_]
    print(f"on the {int(bs_.sum())} violators there: mean |err| split {float(ea.mean()):.3e}  one-pass {float(ec.mean()):.3e}; "
          f"split closer in {int((ea < ec).sum())}, one-pass closer in {int((ec < ea).sum())}; "
          f"max diff / (2^-24 * sum|products|*scale) = {float((diff[sub][bs_].double() / (2.0 ** -24 * mag[bs_].double())).max()):.2f}")
