#!/usr/bin/env python3
"""The kernels of the step boundary, alone and with cold caches: gradient norm, AdamW, the zeroing of the gradient buffer and
the look-ahead packing of every panel of the benchmark model (f32 read + 16-bit written).  Each measurement is preceded by a
1 GiB fill, so nothing is served from the 256 MB memory-side cache as it would be in a tight loop over one kernel.

    python tools/bench_boundary.py [--dtype bf16|f16]
"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_convlstm_amd as U   # noqa: E402
from unet_convlstm_amd import ops   # noqa: E402

L = U._lib
ap = argparse.ArgumentParser()
ap.add_argument("--dtype", default="bf16")
ap.add_argument("--iters", type=int, default=7)
a = ap.parse_args()
dev = torch.device("cuda:0")
if a.dtype == "f16":
    U.set_compute_dtype(torch.float16)
torch.manual_seed(0)
model = U.TemporalUNetDualView(1, 1, base_ch=64, lstm_layers=1, use_skip_lstm=True, use_attention=False).to(dev).train()
opt = U.FusedAdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, max_grad_norm=1.0, loss_scale=2.0 ** 14 if a.dtype == "f16" else None)
data = U.SyntheticSequences(4, 4, 64, 64, seed=1, kind="uniform", device=dev)
for _ in range(3):
    U.train_step(model, opt, data.x, data.y, None, False, None)
torch.cuda.synchronize()
batch = ops._PACK_BATCH
assert batch is not None, "look-ahead packing did not build its batch"
flush = torch.empty(1 << 28, dtype=torch.float32, device=dev)
st = torch.cuda.current_stream()


def timed(fn):
    best = []
    for _ in range(a.iters):
        flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(st)
        fn()
        e1.record(st)
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3)
    best.sort()
    return best[len(best) // 2]


n = int(opt.flat.numel)
print(f"{n} parameters")
fam_names = {0: "generic", 1: "rows<9>", 2: "rows<4>", 3: "transposed<9>", 4: "transposed<4>"}
base = batch.table.data_ptr()
jobs = (L.PackJob * (batch.table.numel() // C.sizeof(L.PackJob))).from_buffer_copy(batch.table.cpu().numpy().tobytes())
tot_t = tot_b = 0.0
for sg, launches in enumerate(batch.segments):
    for dtype, fam, off, nj, blocks in launches:
        first = off // C.sizeof(L.PackJob)
        nbytes = sum(jobs[i].d.N * jobs[i].d.Ktot * 2 for i in range(first, first + nj))
        rbytes = 2 * nbytes          # f32 source elements that land in the panel (padding columns read nothing)
        t = timed(lambda: L.check(L.kernels(dtype).uclstm_pack_weights_batched(C.c_void_p(base + off), nj, fam, blocks, C.c_void_p(st.cuda_stream)), "pack"))
        tot_t += t
        tot_b += nbytes + rbytes
        print(f"segment {sg} {fam_names[fam]:14s} {nj:3d} panels {blocks:6d} blocks  {(nbytes + rbytes) / 1e6:8.1f} MB  {t:8.1f} us  {(nbytes + rbytes) / t / 1e3:7.1f} GB/s")
print(f"packing: {tot_t:.1f} us, {tot_b / 1e6:.1f} MB, {tot_b / tot_t / 1e3:.1f} GB/s")

t = timed(lambda: batch.launch(st))
print(f"packing, all segments back to back: {t:.1f} us ({tot_b / t / 1e3:.1f} GB/s)")

g = opt.flat.flat_g
ss = torch.zeros(1, dtype=torch.float64, device=dev)
t = timed(lambda: L.check(L.lib.uclstm_sumsq(g.data_ptr(), n, ss.data_ptr(), C.c_void_p(st.cuda_stream)), "sumsq"))
print(f"sumsq: {t:.1f} us, {4 * n / t / 1e3:.1f} GB/s")
t = timed(lambda: opt.step())
print(f"optimiser step (sumsq + adamw): {t:.1f} us, {32 * n / t / 1e3:.1f} GB/s")
t = timed(lambda: g.zero_())
print(f"zero the gradient buffer: {t:.1f} us, {4 * n / t / 1e3:.1f} GB/s")

# weight-gradient unpack (f32 panel slabs -> f32 gradient in the reference layout) of the three ConvLSTM weights, slab counts as in
# the benchmark step's range plan
for name, Hd, nslab in (("temporal", 1024, 1), ("skip3", 512, 5), ("skip2", 256, 7)):
    d = ops.lstm_wgrad_unpack_desc(Hd, Hd)
    slabs = torch.randn((nslab, d.N, d.Ktot), device=dev)
    grad = torch.zeros((4 * Hd, 2 * Hd, 3, 3), device=dev)
    ns, sl = ops._slabs_of(slabs)
    t = timed(lambda: L.check(L.lib.uclstm_unpack_wgrad(C.byref(d), slabs.data_ptr(), ns, sl, grad.data_ptr(), 0, C.c_void_p(st.cuda_stream)), "unpack"))
    nb = 4 * d.N * d.Ktot * nslab + 4 * grad.numel()
    print(f"unpack {name} ({nslab} slabs): {t:.1f} us, {nb / t / 1e3:.1f} GB/s")
