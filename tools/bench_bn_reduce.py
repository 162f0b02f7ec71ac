"""Time uclstm_bn_bwd_reduce alone at the headline stage shapes (HIP events, 30 launches after 5), for the A/B of its block count.

The block count is read once per process from UCLSTM_BN_BWD_BLOCKS (csrc/pointwise.hip: bn_bwd_blocks_per_group), so run one process
per value:  for b in 512 1024 2048 4096 8192; do UCLSTM_BN_BWD_BLOCKS=$b python tools/bench_bn_reduce.py; done
Reference points on the same chip: tools/probes/bw_read.hip (two read streams: 5.5 - 6.2 TB/s).
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import unet_convlstm_amd as U  # noqa: E402,F401
from unet_convlstm_amd import _lib as L  # noqa: E402


def p(t):
    return C.c_void_p(t.data_ptr())


def main():
    dev = torch.device("cuda:0")
    # (frames x H x W, groups = timesteps, channels): the four encoder / decoder resolutions of the 64 x 64 seq-20 B=32 step
    shapes = [(640 * 64 * 64, 20, 64), (640 * 32 * 32, 20, 128), (640 * 16 * 16, 20, 256), (640 * 8 * 8, 20, 512)]
    tag = os.environ.get("UCLSTM_BN_BWD_BLOCKS", "default(1024)")
    for pixels, groups, Cp in shapes:
        ppg = pixels // groups
        z = torch.randn(pixels, Cp, device=dev).to(torch.bfloat16)
        da = torch.randn(pixels, Cp, device=dev).to(torch.bfloat16)
        par = [torch.rand(groups, Cp, device=dev) + 0.5 for _ in range(4)]
        rows = int(L.lib.uclstm_bn_bwd_reduce_rows(pixels, ppg))
        partials = torch.empty(rows, Cp, 2, device=dev)
        sums = torch.empty(groups, Cp, 2, device=dev)
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

        def launch():
            L.check(L.lib.uclstm_bn_bwd_reduce(p(z), p(da), p(par[0]), p(par[1]), p(par[2]), p(par[3]), p(partials), p(sums), pixels, ppg,
                                               Cp, st), "bn_bwd_reduce")

        for _ in range(5):
            launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30):
            launch()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 30
        nbytes = 2 * pixels * Cp * 2
        print(f"blocks {tag:>14}  pixels {pixels:>8} C {Cp:>3}  rows {rows:>5}: {ms * 1e3:8.1f} us  {nbytes / ms / 1e6:8.1f} GB/s "
              f"(reduce + sum kernels, {nbytes / 1e6:.0f} MB)", flush=True)


if __name__ == "__main__":
    main()
