#!/usr/bin/env python3
"""Soak run (developer tool, GPU only): N training steps of the benchmark model on one fixed synthetic batch; prints the loss
every few steps and fails on a non-finite value or if the loss does not fall.

    python tools/soak.py [--steps 40] [--batch 32] [--seq 20] [--size 64]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_convlstm_amd as U  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--seq", type=int, default=20)
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f16"])
    ap.add_argument("--attention", action="store_true")
    a = ap.parse_args()
    if a.dtype == "f16":
        U.set_compute_dtype(torch.float16)
    dev = torch.device("cuda", 0)
    torch.manual_seed(7)
    model = U.TemporalUNetDualView(1, 1, base_ch=64, lstm_layers=1, use_skip_lstm=True, use_attention=a.attention).to(dev).train()
    opt = U.FusedAdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, max_grad_norm=1.0, loss_scale=2.0 ** 16 if a.dtype == "f16" else None)
    data = U.SyntheticSequences(a.batch, a.seq, a.size, a.size, seed=3, kind="blobs", device=dev)
    losses = []
    for i in range(a.steps):
        loss, _ = U.train_step(model, opt, data.x, data.y, None, False, None)
        if i % 5 == 0 or i == a.steps - 1:
            losses.append(float(loss))
            print(f"step {i:3d}  loss {losses[-1]:.5f}  grad-norm {float(opt.grad_norm()):.4f}"
                  + (f"  loss-scale {float(opt.scale_state[0]):.0f} good-steps {int(opt.scale_state[2])}" if opt.scale_state is not None else ""),
                  flush=True)
            if not torch.isfinite(loss):
                raise SystemExit("non-finite loss")
    for p in model.parameters():
        if not bool(torch.isfinite(p).all()):
            raise SystemExit("non-finite parameter")
    if not losses[-1] < 0.7 * losses[0]:
        raise SystemExit(f"loss did not fall: {losses[0]} -> {losses[-1]}")
    print("soak ok")


if __name__ == "__main__":
    main()
