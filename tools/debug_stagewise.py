#!/usr/bin/env python3
"""Stage-by-stage eval-mode comparison of the HIP path with the CPU oracle on a seeded fixture case (debug aid).
    python tools/debug_stagewise.py ref_512
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import seeded_case, rel_l2   # noqa: E402
import unet_convlstm_amd as U   # noqa: E402
from oracle import unet_oracle as O   # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "ref_512"
g, sd, x, y, mask, cfg = seeded_case(name)
torch.set_num_threads(16)
m = U.TemporalUNetDualView(1, 1, base_ch=cfg["base_ch"], use_skip_lstm=cfg["skip"]).cuda().eval()
m.load_state_dict(sd)
with torch.no_grad():
    xb, sk = m.encode_once(x[:, 0].cuda())
    rb, rsk = O.encode_once(x[:, 0], sd, False, None, False)
    for nm, a, b in [("x0", sk[3], rsk[3]), ("x1", sk[2], rsk[2]), ("x2", sk[1], rsk[1]), ("x3", sk[0], rsk[0]), ("xb", xb, rb)]:
        a = a.cpu()
        d = (a - b)
        e = rel_l2(a, b)
        # where is the error: per-row and per-column energy
        er = d.pow(2).sum(dim=(0, 1, 3))
        ec = d.pow(2).sum(dim=(0, 1, 2))
        ech = d.pow(2).sum(dim=(0, 2, 3))
        print(f"{nm}: shape {tuple(a.shape)} rel-L2 {e:.3e}; worst rows {er.topk(min(4, er.numel())).indices.tolist()} "
              f"({float(er.max() / er.sum()):.3f} of the error), worst cols {ec.topk(min(4, ec.numel())).indices.tolist()} "
              f"({float(ec.max() / ec.sum()):.3f}), worst channel {int(ech.argmax())} ({float(ech.max() / ech.sum()):.3f})")
    outs, _ = m(x.cuda())
    ro, _ = O.model_forward(sd, x, None, False)
    for t in range(x.shape[1]):
        a, b = outs[t].cpu(), ro[t]
        d = a - b
        er = d.pow(2).sum(dim=(0, 1, 3))
        ec = d.pow(2).sum(dim=(0, 1, 2))
        print(f"out t={t}: rel-L2 {rel_l2(a, b):.3e}; worst rows {er.topk(4).indices.tolist()} ({float(er.max() / er.sum()):.3f}), "
              f"worst cols {ec.topk(4).indices.tolist()} ({float(ec.max() / ec.sum()):.3f}); |ref| mean {float(b.abs().mean()):.4f} "
              f"|diff| mean {float(d.abs().mean()):.5f}")
