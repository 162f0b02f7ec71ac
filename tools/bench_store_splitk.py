#!/usr/bin/env python3
"""Would splitting K help the conv launches whose tile count does not fill whole rounds of 256 CUs?  (developer tool, GPU only)

For each shape of the headline step with 160 / 320 / 640 tiles of 128 x 256 (and one control with exact rounds): the store-epilogue
launch as the step runs it (no BatchNorm statistics here) against `ksplit` K ranges into f32 slabs + uclstm_splitk_finish (bias, bf16
store).  Prints us and TFLOP/s per variant; the split figures INCLUDE the finish pass.
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_convlstm_amd as U   # noqa: E402,F401
from unet_convlstm_amd import _lib as L, ops  # noqa: E402

DEV = "cuda"


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


# (launches per step, images, H = W, C_in, N) -- from `bench.py --dump-launches` (profiles/round3_launches_final.txt)
SHAPES = [(1, 640, 4, 4096, 1024), (6, 640, 8, 512, 512), (2, 640, 4, 1024, 1024), (1, 640, 8, 2048, 512), (1, 640, 4, 512, 1024),
          (1, 640, 4, 1024, 512), (1, 640, 8, 512, 256), (1, 640, 8, 256, 512), (6, 640, 16, 256, 256),
          (1, 48, 16, 4096, 1024)]          # last: the same layer at 256 x 256 seq-12 B=4 (384 tiles = 1.5 rounds)
for per_step, n_img, H, Ci, N in SHAPES:
    pixels = n_img * H * H
    x = (torch.randn(n_img, H, H, Ci, device=DEV) * 0.5).to(torch.bfloat16)
    wp = (torch.randn(N, 9 * Ci, device=DEV) * 0.02).to(torch.bfloat16)
    bias = torch.zeros(N, device=DEV)
    out = torch.empty(n_img, H, H, N, device=DEV, dtype=torch.bfloat16)
    fl = 2.0 * pixels * N * 9 * Ci
    tiles = ((pixels + 127) // 128) * ((N + 255) // 256)
    ms = timeit(lambda: ops.igemm_store([ops.SrcView(x)], wp, (H, H), n_img, [(out, 0, N, 0)], ktap=3, pad=1, bias=bias))
    ref = out.float().clone()
    line = [f"x{per_step} M={pixels:6d} N={N:4d} K={9 * Ci:5d} tiles={tiles:4d} ({tiles / 256:4.2f} rounds)  store {ms * 1e3:6.1f}us/{fl / ms / 1e9:5.0f}"]
    best = (ms, 1)
    for ks in (2, 3, 4, 5, 6, 8):
        nsl = ops.ksplit_used(9 * Ci, ks, 3)
        pre = torch.empty(nsl, pixels, N, device=DEV)

        def split():
            ops.igemm_atomic([ops.SrcView(x)], wp, (H, H), n_img, pre, ks, ktap=3, pad=1, slabs=True)
            L.check(ops._k(wp).uclstm_splitk_finish(ops._p(pre), nsl, pre.stride(0), N, ops._p(bias), None, None, 0, ops._p(out), pixels, N,
                                                    ops._stream()), "splitk_finish")
        t = timeit(split)
        err = float((out.float() - ref).abs().max() / ref.abs().max())
        line.append(f"ks{ks}({nsl * tiles / 256:4.2f}r):{t * 1e3:6.1f}us/{fl / t / 1e9:5.0f} e{err:.0e}")
        if t < best[0]:
            best = (t, ks)
        del pre
    line.append(f"| best ks{best[1]}: {(ms - best[0]) * 1e3 * per_step:+6.1f} us per step saved")
    print("  ".join(line), flush=True)
