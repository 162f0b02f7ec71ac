#!/usr/bin/env python3
"""Per-shape timing of the MFMA kernels at the benchmark configuration (developer tool, GPU only).

    python tools/bench_igemm.py [--iters 5] [--only fwd|lstm|wgrad]

Shapes are the convolutions of TemporalUNetDualView(base_ch=64, skip LSTMs) at 64x64 with T*B = 640 images
(main.py:215-228).  Prints ms and algorithmic TFLOP/s per launch.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_convlstm_amd as U   # noqa: E402
from unet_convlstm_amd import ops  # noqa: E402

DEV = "cuda"
NIMG = 640

CONVS = [  # name, H, C0, C1, Co
    ("inc.3", 64, 64, 0, 64),
    ("down1.0", 32, 64, 0, 128), ("down1.3", 32, 128, 0, 128),
    ("down2.0", 16, 128, 0, 256), ("down2.3", 16, 256, 0, 256),
    ("down3.0", 8, 256, 0, 512), ("down3.3", 8, 512, 0, 512),
    ("bott.0", 4, 512, 0, 1024), ("bott.3", 4, 1024, 0, 1024),
    ("up3.c0", 8, 512, 512, 512), ("up2.c0", 16, 256, 256, 256),
    ("up1.c0", 32, 128, 128, 128), ("up0.c0", 64, 64, 64, 64),
]
LSTMS = [("temporal", 4, 1024), ("skip3", 8, 512), ("skip2", 16, 256)]


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def rnd(*shape):
    return (torch.randn(*shape, device=DEV) * 0.5).to(torch.bfloat16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--shape", default="", help="restrict to one named shape (e.g. down2.3, temporal)")
    a = ap.parse_args()
    global CONVS, LSTMS
    if a.shape:
        CONVS = [c for c in CONVS if c[0] == a.shape]
        LSTMS = [l for l in LSTMS if l[0] == a.shape]
    tot_ms = tot_fl = 0.0
    if a.only in ("", "fwd"):
        for name, H, C0, C1, Co in CONVS:
            Ci = C0 + C1
            srcs = [ops.SrcView(rnd(NIMG, H, H, C0))] + ([ops.SrcView(rnd(NIMG, H, H, C1))] if C1 else [])
            pd = ops.conv_pack_desc(Co, Ci, [C0] + ([C1] if C1 else []), [C0] + ([C1] if C1 else []))
            wp = ops.pack_weights(pd, torch.randn(Co, Ci, 3, 3, device=DEV) * 0.05)
            out = torch.empty(NIMG, H, H, Co, dtype=torch.bfloat16, device=DEV)
            tpg = U._lib.lib.uclstm_igemm_tiles_per_group(NIMG, H, H, 20, Co)
            stats = torch.empty(20, tpg, Co, 2, device=DEV)
            ms = timeit(lambda: ops.igemm_store(srcs, wp, (H, H), NIMG, [(out, 0, Co, 0, 1, 0, 0)], ktap=3, pad=1, groups=20, stats=(None if os.environ.get("NOSTATS") else stats)),
                        a.iters)
            fl = 2.0 * NIMG * H * H * Co * 9 * Ci
            tot_ms += ms
            tot_fl += fl
            print(f"fwd   {name:9s} M={NIMG * H * H:8d} N={Co:5d} K={9 * Ci:6d}  {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s", flush=True)
    if a.only in ("", "lstm"):
        for name, H, Hd in LSTMS:
            B = 32
            x, h = rnd(B, H, H, Hd), rnd(B, H, H, Hd)
            c = torch.randn(B, H, H, Hd, device=DEV)
            pd = ops.lstm_pack_desc(Hd, Hd)
            wp = ops.pack_weights(pd, torch.randn(4 * Hd, 2 * Hd, 3, 3, device=DEV) * 0.02)
            bp = ops.pack_bias(pd, torch.zeros(4 * Hd, device=DEV))
            co, ho = torch.empty_like(c), torch.empty_like(h)
            gates = torch.empty(B, H, H, 4, Hd, dtype=torch.bfloat16, device=DEV)
            ms = timeit(lambda: ops.igemm_lstm(x, h, wp, bp, c, co, ho, gates), a.iters)
            fl = 2.0 * B * H * H * 4 * Hd * 9 * 2 * Hd
            tot_ms += ms * 20
            tot_fl += fl * 20
            print(f"lstm  {name:9s} M={B * H * H:8d} N={4 * Hd:5d} K={18 * Hd:6d}  {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s", flush=True)
    if a.only in ("", "wgrad"):
        for name, H, C0, C1, Co in CONVS:
            Ci = C0 + C1
            srcs = [ops.SrcView(rnd(NIMG, H, H, C0))] + ([ops.SrcView(rnd(NIMG, H, H, C1))] if C1 else [])
            pd = ops.conv_pack_desc(Co, Ci, [C0] + ([C1] if C1 else []), [C0] + ([C1] if C1 else []))
            dy = rnd(NIMG, H, H, Co)
            ms = timeit(lambda: ops.igemm_wgrad(srcs, [(dy, 0, Co, 0, 1, 0, 0)], pd.N, pd.Ktot, (H, H), NIMG, ktap=3, pad=1), a.iters)
            fl = 2.0 * NIMG * H * H * Co * 9 * Ci
            tot_ms += ms
            tot_fl += fl
            print(f"wgrad {name:9s} M={NIMG * H * H:8d} N={Co:5d} K={9 * Ci:6d}  {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s", flush=True)
        for name, H, Hd in LSTMS:
            T, B = 20, 32
            x, hp = rnd(T * B, H, H, Hd), rnd(T * B, H, H, Hd)
            dg = rnd(T * B, H, H, 4 * Hd)
            ud = ops.lstm_wgrad_unpack_desc(Hd, Hd)
            ms = timeit(lambda: ops.igemm_wgrad([ops.SrcView(x), ops.SrcView(hp)], [(dg, 0, 4 * Hd, 0, 1, 0, 0)], ud.N, ud.Ktot, (H, H),
                                                T * B, ktap=3, pad=1), max(1, a.iters // 2))
            fl = 2.0 * T * B * H * H * 4 * Hd * 9 * 2 * Hd
            tot_ms += ms
            tot_fl += fl
            print(f"wgrad lstm.{name:5s} M={T * B * H * H:8d} N={4 * Hd:5d} K={18 * Hd:6d}  {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s", flush=True)
    print(f"total {tot_ms:.2f} ms, {tot_fl / tot_ms / 1e9:.1f} TFLOP/s weighted")


if __name__ == "__main__":
    main()
