"""Host-side operator layer over the C ABI (include/uclstm.h).

Internal activation format: ``torch.bfloat16`` tensor ``[N, H, W, Cp]`` (NHWC, ``Cp`` = channels
padded to a multiple of 8, padding channels are exactly zero); cell state is ``float32`` in the
same layout.  PyTorch is used here only for device memory, the current HIP stream and autograd
bookkeeping -- every arithmetic kernel is in libuclstm.so.  Nothing in this file runs on CPU
tensors: ``_dev`` raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence, Tuple

import torch

from . import _lib as L

BF16 = torch.bfloat16
F16 = torch.float16
F32 = torch.float32
ACT = (BF16, F16)          # the 16-bit activation / panel dtypes; every operator takes its kernel set from its input's dtype

# Compute dtype of NEW activation tensors at the module boundary (f32 NCHW in -> 16-bit NHWC): bfloat16 by default,
# torch.float16 = the fp16-MFMA twin kernels (BASELINE.json configs[3]).  Downstream operators follow their input's dtype.
_COMPUTE_DTYPE = BF16


def set_compute_dtype(dtype) -> None:
    global _COMPUTE_DTYPE
    if dtype not in ACT:
        raise L.UclstmError(f"compute dtype must be torch.bfloat16 or torch.float16, got {dtype}")
    _COMPUTE_DTYPE = dtype


def get_compute_dtype():
    return _COMPUTE_DTYPE


class compute_dtype:
    """``with ops.compute_dtype(torch.float16): out, _ = model(x)`` -- the forward pass inside runs on the fp16 kernels (the
    backward pass follows the dtype of the saved activations, wherever it runs)."""

    def __init__(self, dtype):
        self.dtype = dtype

    def __enter__(self):
        self.prev = _COMPUTE_DTYPE
        set_compute_dtype(self.dtype)

    def __exit__(self, *exc):
        set_compute_dtype(self.prev)


def _k(t: torch.Tensor):
    """Kernel set (bf16 / fp16 twins) for a 16-bit activation tensor."""
    return L.kernels(t.dtype)


def cpad(c: int) -> int:
    return (c + 7) // 8 * 8


def kseg(c: int) -> int:
    return (c + 63) // 64 * 64


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t: torch.Tensor, dtype=None, what: str = "tensor") -> torch.Tensor:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise L.UclstmError(f"{what}: a HIP device tensor is required (this package has no CPU path)")
    if dtype is not None and (t.dtype not in dtype if isinstance(dtype, tuple) else t.dtype != dtype):
        raise L.UclstmError(f"{what}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise L.UclstmError(f"{what}: must be contiguous")
    return t


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


# Optional per-launch timing of the MFMA kernels (bench.py's roofline leg): when PROFILE is a list every GEMM launch
# is bracketed by HIP events on the launch stream and (kernel, algorithmic flops, start, stop) is appended.
PROFILE: Optional[list] = None


# Optional record of which forward-family kernel each launch takes (uclstm_igemm_fwd_shape: 0 / 1 per-tap shapes, 2 patch loop,
# 3 ring kernel): tests set it to a list to assert that a parity case exercised the kernel it is meant for.
SHAPE_LOG: Optional[list] = None
# Same, as (epilogue, shape) pairs -- epilogue 0 store, 1 fused ConvLSTM cell, 2 split-K partial tiles: the BASELINE-shape
# parity tests assert that the fused-cell, split-K and patch kernels the benchmark runs were the ones compared.
KERNEL_LOG: Optional[list] = None


# Same again with the geometry, for the full-size property tests: ("fwd", epilogue, shape, image width, N), ("wgrad", shape,
# N, Ktot, pixel-range slabs) and ("split", what, image ranges) for a launch cut at the 2-GiB descriptor range.
LAUNCH_LOG: Optional[list] = None


def _log_shape(d) -> None:
    if SHAPE_LOG is not None or KERNEL_LOG is not None or LAUNCH_LOG is not None:
        shp = int(L.lib.uclstm_igemm_fwd_shape(C.byref(d)))
        if SHAPE_LOG is not None:
            SHAPE_LOG.append(shp)
        if KERNEL_LOG is not None:
            KERNEL_LOG.append((int(d.epi), shp))
        if LAUNCH_LOG is not None:
            LAUNCH_LOG.append(("fwd", int(d.epi), shp, int(d.W), int(d.N)))


# Same for the HBM-bound kernels (BatchNorm passes, pooling, LSTM point-wise): (kernel, algorithmic BYTES, start, stop, note).
PROFILE_HBM: Optional[list] = None


def _timed_hbm(kind: str, nbytes: float, launch) -> None:
    if PROFILE_HBM is None:
        launch()
        return
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    PROFILE_HBM.append((kind, nbytes, e0, e1, ""))


_SHAPE_NAMES = {0: "pertap128x128", 1: "pertap64x256", 2: "patch128x256", 3: "ring64"}


def _kernel_kind(kind: str, d) -> str:
    """Profile key of a forward-family launch: epilogue family + the kernel the library picks for this descriptor (the
    rocprofv3 name is igemm_fwd_kernel<epilogue, shape, sources> / igemm_fwd_c64_kernel), so that bench.py's roofline leg
    prices each KERNEL, not a mix of MFMA-bound 3x3 launches and HBM-bound 1x1 / 2x2 ones."""
    if PROFILE is None:
        return kind
    return f"{kind}[{_SHAPE_NAMES.get(int(L.lib.uclstm_igemm_fwd_shape(C.byref(d))), '?')}]"


def _timed(kind: str, flops: float, launch, note: str = "", nbytes: float = 0.0) -> None:
    """``nbytes``: ALGORITHMIC HBM bytes of the launch -- every operand read once, the result written once (split-K /
    pixel-range slabs counted as ONE result) -- the partner of the PMC traffic figure in bench.py's roofline leg."""
    if PROFILE is None:
        launch()
        return
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    PROFILE.append((kind, flops, e0, e1, note, float(nbytes)))


def _nb(*tensors) -> int:
    return sum(t.numel() * t.element_size() for t in tensors if t is not None)


# ---------------------------------------------------------------------------------------------
# descriptors
# ---------------------------------------------------------------------------------------------
class SrcView:
    """One input of a convolution: NHWC bf16 tensor placed at (offY, offX) in the output frame."""

    def __init__(self, t: torch.Tensor, offY: int = 0, offX: int = 0):
        _dev(t, ACT, "conv source")
        assert t.dim() == 4 and t.shape[3] % 8 == 0
        if offY < 0 or offX < 0:
            # measured: a source that overhangs the output frame reads out of range in the input-gradient / weight-gradient
            # kernels (garbage, not zeros).  modules.Up crops such a source before it gets here.
            raise L.UclstmError("conv source offsets must be >= 0 (a source is placed INSIDE the output frame)")
        self.t, self.offY, self.offX = t, offY, offX

    def fill(self, s: L.Src):
        s.ptr = self.t.data_ptr()
        s.C = self.t.shape[3]
        s.Hs, s.Ws = self.t.shape[1], self.t.shape[2]
        s.offY, s.offX = self.offY, self.offX


def _fill_seg(sg: L.Seg, t: torch.Tensor, n_begin: int, n_end: int, c_off: int = 0, scale: int = 1, oy: int = 0, ox: int = 0):
    sg.ptr = t.data_ptr()
    sg.n_begin, sg.n_end = n_begin, n_end
    sg.C, sg.c_off = t.shape[3], c_off
    sg.Hd, sg.Wd = t.shape[1], t.shape[2]
    sg.scale, sg.oy, sg.ox = scale, oy, ox


def conv_pack_desc(Co: int, Ci_total: int, c_valid: Sequence[int], c_pad: Sequence[int]) -> L.PackDesc:
    """Forward panel of a 3x3 conv whose input is the channel concat of the sources (train/unet.py:70,:98)."""
    d = L.PackDesc()
    d.N, d.taps, d.nsrc = cpad(Co), 9, len(c_valid)
    for i in range(2):
        d.kseg[i] = kseg(c_pad[i]) if i < len(c_valid) else 0
        d.cvalid[i] = c_valid[i] if i < len(c_valid) else 0
    d.choff[0], d.choff[1] = 0, c_valid[0]
    d.Ktot = 9 * (d.kseg[0] + d.kseg[1])
    d.n_mode, d.n_valid, d.n_cp = L.NMODE_IDENTITY, Co, 0
    d.k_mode, d.k_hdp, d.k_hd, d.tap_flip = L.KMODE_IDENTITY, 0, 0, 0
    d.stride_n, d.stride_k, d.stride_tap, d.stride_ntap = Ci_total * 9, 9, 1, 0
    return d


def conv_dgrad_pack_desc(Co: int, Ci_total: int, c_valid_s: int) -> L.PackDesc:
    """Input-gradient panel for ONE source: rows = its input channels, K = (flipped tap, out channel)."""
    d = L.PackDesc()
    d.N, d.taps, d.nsrc = cpad(c_valid_s), 9, 1
    d.kseg[0], d.kseg[1] = kseg(cpad(Co)), 0
    d.cvalid[0], d.cvalid[1] = Co, 0
    d.choff[0], d.choff[1] = 0, 0
    d.Ktot = 9 * d.kseg[0]
    d.n_mode, d.n_valid, d.n_cp = L.NMODE_IDENTITY, c_valid_s, 0
    d.k_mode, d.k_hdp, d.k_hd, d.tap_flip = L.KMODE_IDENTITY, 0, 0, 1
    d.stride_n, d.stride_k, d.stride_tap, d.stride_ntap = 9, Ci_total * 9, 1, 0
    return d


def im2col_pack_desc(Co: int, Ci: int, Kp: int) -> L.PackDesc:
    d = L.PackDesc()
    d.N, d.taps, d.nsrc = cpad(Co), 1, 1
    d.kseg[0], d.kseg[1] = kseg(Kp), 0
    d.cvalid[0], d.cvalid[1] = 9 * Ci, 0
    d.choff[0] = d.choff[1] = 0
    d.Ktot = d.kseg[0]
    d.n_mode, d.n_valid, d.n_cp = L.NMODE_IDENTITY, Co, 0
    d.k_mode, d.k_hdp, d.k_hd, d.tap_flip = L.KMODE_IM2COL, 9, Ci, 0
    d.stride_n, d.stride_k, d.stride_tap, d.stride_ntap = Ci * 9, 9, 1, 0
    return d


def lstm_pack_desc(Hd: int, Cx: int, ksize: int = 3) -> L.PackDesc:
    """Gate-interleaved forward panel of the ConvLSTM gate conv [4Hd, Cx+Hd, k, k] (train/unet.py:19)."""
    taps = ksize * ksize
    d = L.PackDesc()
    d.N, d.taps, d.nsrc = 64 * ((Hd + 15) // 16), taps, 2
    d.kseg[0], d.kseg[1] = kseg(cpad(Cx)), kseg(cpad(Hd))
    d.cvalid[0], d.cvalid[1] = Cx, Hd
    d.choff[0], d.choff[1] = 0, Cx
    d.Ktot = taps * (d.kseg[0] + d.kseg[1])
    d.n_mode, d.n_valid, d.n_cp = L.NMODE_LSTM, Hd, 0
    d.k_mode, d.k_hdp, d.k_hd, d.tap_flip = L.KMODE_IDENTITY, 0, 0, 0
    d.stride_n, d.stride_k, d.stride_tap, d.stride_ntap = (Cx + Hd) * taps, taps, 1, 0
    return d


def lstm_half_pack_desc(Hd: int, Cx: int, half: str, ksize: int = 3) -> L.PackDesc:
    """One half of the gate convolution as its own gate-interleaved panel: ``half='x'`` = W_x (input channels [0, Cx) of
    train/unet.py:19's conv, applied to x_t), ``'h'`` = W_h (channels [Cx, Cx+Hd), applied to h_{t-1}).  W_x * x_t does not
    depend on the recurrence, so it is hoisted out of the time loop of train/unet.py:55-57 and computed for all timesteps
    by one GEMM; the serial step then carries only W_h * h_{t-1} (half the K, half the weight traffic per step)."""
    taps = ksize * ksize
    cs = Cx if half == "x" else Hd
    d = L.PackDesc()
    d.N, d.taps, d.nsrc = 64 * ((Hd + 15) // 16), taps, 1
    d.kseg[0], d.kseg[1] = kseg(cpad(cs)), 0
    d.cvalid[0], d.cvalid[1] = cs, 0
    d.choff[0], d.choff[1] = (0 if half == "x" else Cx), 0
    d.Ktot = taps * d.kseg[0]
    d.n_mode, d.n_valid, d.n_cp = L.NMODE_LSTM, Hd, 0
    d.k_mode, d.k_hdp, d.k_hd, d.tap_flip = L.KMODE_IDENTITY, 0, 0, 0
    d.stride_n, d.stride_k, d.stride_tap, d.stride_ntap = (Cx + Hd) * taps, taps, 1, 0
    return d


def lstm_dgrad_pack_desc(Hd: int, Cx: int, c_valid_s: int, ksize: int = 3) -> L.PackDesc:
    """Input-gradient panel of the gate conv for one source (x or h); K = (flipped tap, gate*Hd_p + hc)."""
    taps = ksize * ksize
    Hdp = cpad(Hd)
    d = L.PackDesc()
    d.N, d.taps, d.nsrc = cpad(c_valid_s), taps, 1
    d.kseg[0], d.kseg[1] = kseg(4 * Hdp), 0
    d.cvalid[0], d.cvalid[1] = 4 * Hd, 0
    d.choff[0] = d.choff[1] = 0
    d.Ktot = taps * d.kseg[0]
    d.n_mode, d.n_valid, d.n_cp = L.NMODE_IDENTITY, c_valid_s, 0
    d.k_mode, d.k_hdp, d.k_hd, d.tap_flip = L.KMODE_GATES, Hdp, Hd, 1
    d.stride_n, d.stride_k, d.stride_tap, d.stride_ntap = taps, (Cx + Hd) * taps, 1, 0
    return d


def lstm_wgrad_unpack_desc(Hd: int, Cx: int, ksize: int = 3) -> L.PackDesc:
    """Panel-gradient layout of the gate conv: rows = gate*Hd_p + hc (the dgates channel order)."""
    taps = ksize * ksize
    Hdp = cpad(Hd)
    d = L.PackDesc()
    d.N, d.taps, d.nsrc = 4 * Hdp, taps, 2
    d.kseg[0], d.kseg[1] = kseg(cpad(Cx)), kseg(Hdp)
    d.cvalid[0], d.cvalid[1] = Cx, Hd
    d.choff[0], d.choff[1] = 0, Cx
    d.Ktot = taps * (d.kseg[0] + d.kseg[1])
    d.n_mode, d.n_valid, d.n_cp = L.NMODE_TAPMAJOR, Hd, Hdp
    d.k_mode, d.k_hdp, d.k_hd, d.tap_flip = L.KMODE_IDENTITY, 0, 0, 0
    d.stride_n, d.stride_k, d.stride_tap = (Cx + Hd) * taps, taps, 1
    d.stride_ntap = Hd * (Cx + Hd) * taps
    return d


def convt_pack_desc(Ci: int, Co: int) -> L.PackDesc:
    """Forward panel of ConvTranspose2d(Ci, Co, 2, stride 2): rows (tap, co), K = ci (weight [Ci,Co,2,2], train/unet.py:90)."""
    Cop = cpad(Co)
    d = L.PackDesc()
    d.N, d.taps, d.nsrc = 4 * Cop, 1, 1
    d.kseg[0], d.kseg[1] = kseg(cpad(Ci)), 0
    d.cvalid[0], d.cvalid[1] = Ci, 0
    d.choff[0] = d.choff[1] = 0
    d.Ktot = d.kseg[0]
    d.n_mode, d.n_valid, d.n_cp = L.NMODE_TAPMAJOR, Co, Cop
    d.k_mode, d.k_hdp, d.k_hd, d.tap_flip = L.KMODE_IDENTITY, 0, 0, 0
    d.stride_n, d.stride_k, d.stride_tap, d.stride_ntap = 4, Co * 4, 0, 1
    return d


def convt_dgrad_pack_desc(Ci: int, Co: int) -> L.PackDesc:
    d = L.PackDesc()
    d.N, d.taps, d.nsrc = cpad(Ci), 4, 1
    d.kseg[0], d.kseg[1] = kseg(cpad(Co)), 0
    d.cvalid[0], d.cvalid[1] = Co, 0
    d.choff[0] = d.choff[1] = 0
    d.Ktot = 4 * d.kseg[0]
    d.n_mode, d.n_valid, d.n_cp = L.NMODE_IDENTITY, Ci, 0
    d.k_mode, d.k_hdp, d.k_hd, d.tap_flip = L.KMODE_IDENTITY, 0, 0, 0
    d.stride_n, d.stride_k, d.stride_tap, d.stride_ntap = Co * 4, 4, 1, 0
    return d


# Inference-time panel cache.  There is no module-global cache: a caller that owns frozen weights (streaming.
# StreamingPredictor) creates a ``PanelCache`` and makes it current around its forward calls; the panels live exactly as long
# as that object (a captured HIP graph has their addresses baked in, so they must not be freed under it).  Entries are
# validated by the weight tensor's version AND by ``WEIGHTS_EPOCH``, which the fused optimiser bumps on every step: it updates
# parameters through raw pointers, which does not change tensor versions.
WEIGHTS_EPOCH = 0


def weights_changed() -> None:
    """Called by code that rewrites parameter storage behind autograd's back (optim.FusedAdamW.step)."""
    global WEIGHTS_EPOCH
    WEIGHTS_EPOCH += 1


class PanelCache:
    """Packed panels of frozen weights, keyed by (storage address, element offset, pack descriptor)."""

    def __init__(self):
        self.entries: dict = {}
        self.epoch = WEIGHTS_EPOCH

    def stale(self) -> bool:
        return self.epoch != WEIGHTS_EPOCH

    def clear(self) -> None:
        self.entries.clear()
        self.epoch = WEIGHTS_EPOCH

    def __enter__(self):
        global _ACTIVE_CACHE
        self._prev = _ACTIVE_CACHE
        _ACTIVE_CACHE = self
        return self

    def __exit__(self, *exc):
        global _ACTIVE_CACHE
        _ACTIVE_CACHE = self._prev


_ACTIVE_CACHE: Optional[PanelCache] = None


# Look-ahead packing for training steps (engine.train_step): the weights only change in the optimiser step, so every panel of
# a step -- the forward panels and, 15 ms later, the input-gradient panels -- can be packed at the START of the step.  The
# list of pack calls of one step is remembered; at the start of the next step all of them are replayed on the side stream
# (HBM-bound 10-80 us kernels, ~1.5 ms per step in a row on the main stream before), each followed by an event, and
# pack_weights() on the main stream turns into a wait for that event.  A call that the plan does not know packs in line.
PREPACK = os.environ.get("UCLSTM_PREPACK", "1") != "0"
_PACK_PLAN: list = []          # [(key, desc copy, weight, elem_offset)] in call order, recorded during the previous step
_PACK_READY: dict = {}         # key -> (panel, event) produced by prepack_begin() for the current step
_PACK_RECORDING = False


PACK_BATCHED = os.environ.get("UCLSTM_PACK_BATCHED", "1") != "0"


PACK_SEGMENTS = tuple(float(x) for x in os.environ.get("UCLSTM_PACK_SEGMENTS", "0.02,0.08,0.2,0.35,0.5,0.65,0.8").split(",") if x)


def pack_segments(sizes: Sequence[int], fractions: Sequence[float]) -> List[int]:
    """Segment index of every panel (in order of first use): panel i belongs to segment s when the bytes BEFORE it have
    reached ``fractions[s - 1]`` of the total (and not yet ``fractions[s]``); non-decreasing, segment 0 is never empty."""
    total, run, seg_of = float(sum(sizes)), 0, []
    for sz in sizes:
        seg_of.append(sum(1 for f in fractions if run >= f * total))
        run += sz
    return seg_of


class _PackBatch:
    """All look-ahead panels of a step as a few launches per kernel family (uclstm_pack_weights_batched).

    The job table (descriptors, weight and panel pointers, block ranges) is built once and lives on the device; the panels
    are persistent, which is safe inside train_step: the step's packing waits for everything the main stream has been
    given (side.wait_stream(main) in prepack_begin), i.e. for the last reader of the previous step.

    The panels are cut, in order of first use, into segments at the cumulative byte fractions PACK_SEGMENTS; a segment is one
    launch per family followed by an event, so the step's first convolutions wait for the first few per cent of the bytes
    only.  (Holding the later segments back until the main stream reaches the MFMA-bound layers was tried and measured no
    different: packing is ~0.55 ms of HBM traffic per step and costs about that wherever it runs, profiles/round2_notes.md.)
    """

    def __init__(self, items):
        self.sig = tuple(key for key, _, _, _ in items)
        self.panels, self.keep, self.segments = {}, [], []            # segments: [[(dtype, fam, byte offset, njobs, blocks)]]
        dev = items[0][2].device
        seg_of = pack_segments([desc.N * desc.Ktot for _, desc, _, _ in items], PACK_SEGMENTS)
        jobs_all = []
        for n, seg in enumerate(sorted(set(seg_of))):
            groups: dict = {}
            for (key, desc, w, off), sg in zip(items, seg_of):
                if sg != seg:
                    continue
                wp = torch.empty((desc.N, desc.Ktot), dtype=key[3], device=dev)
                job = L.PackJob()
                fam = L.lib.uclstm_pack_job_init(C.byref(job), C.byref(desc), C.c_void_p(w.data_ptr() + 4 * off), _p(wp), 0)
                L.check(min(fam, 0), "pack_job_init")
                groups.setdefault((key[3], fam), []).append(job)
                self.panels[key] = (wp, n)
                self.keep.append(w)
            launches = []
            for (dtype, fam), jobs in groups.items():
                b0, first = 0, len(jobs_all)
                for j in jobs:
                    j.block0 = b0
                    b0 += j.nblocks
                    jobs_all.append(j)
                launches.append((dtype, fam, first * C.sizeof(L.PackJob), len(jobs), b0))
            self.segments.append(launches)
        arr = (L.PackJob * len(jobs_all))(*jobs_all)
        self.table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self.events: list = [None] * len(self.segments)

    def launch(self, side) -> None:
        """Launch every segment on ``side``; events[s] is recorded behind segment s."""
        base, st = self.table.data_ptr(), C.c_void_p(side.cuda_stream)
        for sg, launches in enumerate(self.segments):
            for dtype, fam, byte_off, n, blocks in launches:
                L.check(L.kernels(dtype).uclstm_pack_weights_batched(C.c_void_p(base + byte_off), n, fam, blocks, st), "pack_weights_batched")
            self.events[sg] = torch.cuda.Event()
            self.events[sg].record(side)

    def panel(self, key, main) -> torch.Tensor:
        wp, sg = self.panels[key]
        main.wait_event(self.events[sg])
        return wp


_PACK_BATCH: Optional[_PackBatch] = None


def prepack_begin() -> None:
    """Start of a training step whose parameters will not change before its backward pass has run."""
    global _PACK_RECORDING, _PACK_PLAN, _PACK_BATCH
    _PACK_READY.clear()
    _PACK_RECORDING = False
    if not PREPACK:
        return
    plan, _PACK_PLAN = _PACK_PLAN, []
    _PACK_RECORDING = True
    if not plan:
        return
    dev = plan[0][2].device
    main, side = torch.cuda.current_stream(dev), side_stream(dev)
    if PACK_BATCHED:
        items, seen = [], set()
        for it in plan:
            if it[0] not in seen and it[2].data_ptr() == it[0][0]:
                seen.add(it[0])
                items.append(it)
        if not items:
            return
        if _PACK_BATCH is None or _PACK_BATCH.sig != tuple(it[0] for it in items):
            _PACK_BATCH = _PackBatch(items)          # panels allocated on the main stream: their later reuse is ordered there
        side.wait_stream(main)                       # the optimiser step that produced these weights + the panels' last readers
        _PACK_BATCH.launch(side)
        for key in _PACK_BATCH.panels:
            _PACK_READY[key] = _PACK_BATCH
        return
    side.wait_stream(main)                       # the optimiser step that produced these weights
    with torch.cuda.stream(side):
        for key, desc, w, off in plan:
            if key in _PACK_READY or w.data_ptr() != key[0]:
                continue
            wp = torch.empty((desc.N, desc.Ktot), dtype=key[3], device=w.device)
            L.check(L.kernels(key[3]).uclstm_pack_weights(C.byref(desc), C.c_void_p(w.data_ptr() + 4 * off), _p(wp), _stream()), "pack_weights")
            wp.record_stream(main)
            ev = torch.cuda.Event()
            ev.record(side)
            _PACK_READY[key] = (wp, ev)


def prepack_end() -> None:
    """End of the step (before the optimiser changes the weights): panels packed ahead are no longer valid."""
    global _PACK_RECORDING
    _PACK_READY.clear()
    _PACK_RECORDING = False


def pack_weights(desc: L.PackDesc, w: torch.Tensor, elem_offset: int = 0, dtype=BF16) -> torch.Tensor:
    """f32 reference-layout weight -> 16-bit panel of ``dtype`` (the dtype of the activations it will multiply)."""
    _dev(w, F32, "weight")
    key = None
    if _PACK_RECORDING:
        key = (w.data_ptr(), elem_offset, bytes(desc), dtype)
        d2 = L.PackDesc()
        C.memmove(C.byref(d2), C.byref(desc), C.sizeof(L.PackDesc))
        _PACK_PLAN.append((key, d2, w, elem_offset))
        hit = _PACK_READY.get(key)
        if hit is not None:
            main = torch.cuda.current_stream(w.device)
            if hit is _PACK_BATCH:
                return hit.panel(key, main)
            main.wait_event(hit[1])
            return hit[0]
        key = None
    cache = _ACTIVE_CACHE
    if cache is not None:
        if cache.stale():
            raise L.UclstmError("PanelCache: the weights changed (optimiser step) while a panel cache was current; its owner "
                                "must drop it first (StreamingPredictor does so in step())")
        key = (w.data_ptr(), elem_offset, bytes(desc), dtype)
        hit = cache.entries.get(key)
        if hit is not None and hit[0] == w._version:
            return hit[1]
    wp = torch.empty((desc.N, desc.Ktot), dtype=dtype, device=w.device)
    L.check(L.kernels(dtype).uclstm_pack_weights(C.byref(desc), C.c_void_p(w.data_ptr() + 4 * elem_offset), _p(wp), _stream()), "pack_weights")
    if key is not None:
        cache.entries[key] = (w._version, wp)
    return wp


def pack_bias(desc: L.PackDesc, b: torch.Tensor) -> torch.Tensor:
    _dev(b, F32, "bias")
    if desc.n_mode == L.NMODE_IDENTITY and desc.N == desc.n_valid == b.numel() and b.is_contiguous() and b.data_ptr() % 16 == 0:
        return b.detach()          # panel-row order == channel order and no padding rows: the bias IS its panel (no kernel)
    cache, key = _ACTIVE_CACHE, None
    if cache is not None and not cache.stale():              # frozen weights (StreamingPredictor): once, like the weight panels
        key = ("bias", b.data_ptr(), bytes(desc))
        hit = cache.entries.get(key)
        if hit is not None and hit[0] == b._version:
            return hit[1]
    bp = torch.empty((desc.N,), dtype=F32, device=b.device)
    L.check(L.lib.uclstm_pack_bias(C.byref(desc), _p(b), _p(bp), _stream()), "pack_bias")
    if key is not None:
        cache.entries[key] = (b._version, bp)
    return bp


def unpack_wgrad(desc: L.PackDesc, dwp: torch.Tensor, like: torch.Tensor) -> torch.Tensor:
    """dwp: f32 panel gradient [N, Ktot] or its per-pixel-range slabs [splits, N, Ktot] (added up here)."""
    grad = torch.empty_like(like, dtype=F32, memory_format=torch.contiguous_format)
    ns, st = _slabs_of(dwp)
    L.check(L.lib.uclstm_unpack_wgrad(C.byref(desc), _p(dwp), ns, st, _p(grad), 0, _stream()), "unpack_wgrad")
    return grad


def _slabs_of(dwp: torch.Tensor) -> Tuple[int, int]:
    return (dwp.shape[0], dwp.stride(0)) if dwp.dim() == 3 else (1, 0)


# ---------------------------------------------------------------------------------------------
# weight gradients on a side stream
# ---------------------------------------------------------------------------------------------
# The weight-gradient GEMM of a layer is not on the critical path of backward (only the optimiser needs it), while the
# HBM-bound kernels between the input-gradient GEMMs (BatchNorm backward, pool, point-wise LSTM, packing) leave the MFMA
# pipes idle.  When the parameter already owns a gradient buffer (``optim.FlatParams`` / ``zero_grad(set_to_none=False)``),
# the weight-gradient GEMM + its unpack-ACCUMULATE into ``weight.grad`` are enqueued on a second HIP stream and the
# autograd function returns ``None`` for that parameter; the main stream re-joins at the end of the backward pass
# (autograd engine callback), before anything can read the gradients.  ``GRAD_SIDE_HOOKS`` lets data-parallel code learn
# that a parameter's gradient has been enqueued (the hook runs with the side stream current).
ASYNC_WGRAD = os.environ.get("UCLSTM_ASYNC_WGRAD", "1") != "0"
BN_RUNNING_ON_SIDE = os.environ.get("UCLSTM_BN_RUNNING_ON_SIDE", "1") != "0"     # running-statistics recursion on the second stream
_FWD_SIDE_PENDING: set = set()


def join_forward_side(device) -> None:
    """Make the current stream wait for forward-pass work that was put on the second stream (BatchNorm running statistics).
    Called where such results become observable: at the end of a module's public forward, before evaluation-mode
    BatchNorm reads the running statistics; the backward pass joins the second stream anyway."""
    key = str(device)
    if key in _FWD_SIDE_PENDING:
        _FWD_SIDE_PENDING.discard(key)
        torch.cuda.current_stream(device).wait_stream(side_stream(device))


PARAM_GRADS_ON_SIDE = os.environ.get("UCLSTM_PARAM_GRADS_ON_SIDE", "1") != "0"   # BatchNorm parameter gradients on the weight-gradient stream
HOIST_X = os.environ.get("UCLSTM_HOIST_X", "0") == "1"               # x half of the ConvLSTM gate conv as one GEMM over all T (off: measured slower)
POOL_SKIP = os.environ.get("UCLSTM_POOL_SKIP", "1") != "0"            # skip-connection gradient added inside max-pool backward
DIRECT_GRADS = os.environ.get("UCLSTM_DIRECT_GRADS", "1") != "0"     # small parameter gradients written by the backward kernels
GRAD_SIDE_HOOKS: list = []
_WGRAD_OVERLAPPED = False      # True while a weight-gradient GEMM is being enqueued on the side stream
_SIDE_STREAMS = {}
_LAUNCH_STREAMS = {}
_SIDE_STREAM_PROBES = {}       # device -> number of candidate streams tried (diagnostics)
_JOIN_PENDING = set()


def _streams_overlap(main: "torch.cuda.Stream", cand: "torch.cuda.Stream", spin_us: int = 300) -> bool:
    """True when a kernel on ``cand`` executes while one on ``main`` is still running.  The HIP runtime multiplexes
    streams onto a few hardware queues; two streams that share one are serialised, and which ones collide depends on
    every stream created before (an RCCL communicator shifts the assignment: measured 37.7 -> 42.0 ms per step with
    the weight gradients silently serialised behind the main stream)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    main.synchronize()
    cand.synchronize()
    e0.record(main)
    L.check(L.lib.uclstm_stream_spin(spin_us, C.c_void_p(main.cuda_stream)), "stream_spin")
    L.check(L.lib.uclstm_stream_spin(spin_us, C.c_void_p(cand.cuda_stream)), "stream_spin")
    e1.record(cand)
    cand.synchronize()
    main.synchronize()
    return e0.elapsed_time(e1) < 1.5e-3 * spin_us


def _pick_stream(device, others) -> tuple:
    """(stream, tried): the first of up to eight fresh streams that demonstrably runs concurrently with every stream in
    ``others``; the first candidate (and a warning) when none does."""
    cands = []
    if os.environ.get("UCLSTM_SIDE_STREAM_PROBE", "1") != "0" and not torch.cuda.is_current_stream_capturing():
        # first launch of the spin kernel outside the timed comparison (code-object load)
        L.check(L.lib.uclstm_stream_spin(1, C.c_void_p(others[0].cuda_stream)), "stream_spin")
        for _ in range(8):
            cands.append(torch.cuda.Stream(device=device))
            if all(_streams_overlap(o, cands[-1]) for o in others):
                return cands[-1], len(cands)
        import warnings
        warnings.warn("unet_convlstm_amd: no HIP stream runs concurrently with the ones in use; work meant to overlap the "
                      "backward pass will be serialised behind it (try GPU_MAX_HW_QUEUES=8)")
    return (cands[0] if cands else torch.cuda.Stream(device=device)), len(cands)


def side_stream(device) -> "torch.cuda.Stream":
    """The second stream of ``device`` (weight gradients).  Chosen once, at first use, as a stream that runs concurrently
    with the stream current at that moment."""
    key = str(device)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key], _SIDE_STREAM_PROBES[key] = _pick_stream(device, [torch.cuda.current_stream(device)])
    return _SIDE_STREAMS[key]


def launch_stream(device, main: "torch.cuda.Stream") -> "torch.cuda.Stream":
    """A third stream that carries no kernels, only waits: data-parallel code enqueues its collectives from it so that
    neither ``main`` nor the side stream ever blocks on the other.  It must not share a hardware queue with either (a
    wait packet would hold back the kernels queued behind it)."""
    key = str(device)
    if key not in _LAUNCH_STREAMS:
        _LAUNCH_STREAMS[key], _ = _pick_stream(device, [main, side_stream(device)])
    return _LAUNCH_STREAMS[key]


def _schedule_join(device) -> None:
    key = str(device)
    if key in _JOIN_PENDING:
        return
    _JOIN_PENDING.add(key)

    def join():
        _JOIN_PENDING.discard(key)
        torch.cuda.current_stream(device).wait_stream(side_stream(device))

    torch.autograd.Variable._execution_engine.queue_callback(join)


def wgrad_into_param(weight: torch.Tensor, desc: L.PackDesc, inputs: Sequence[torch.Tensor], run_gemm) -> Optional[torch.Tensor]:
    """Weight gradient of one layer.  ``run_gemm()`` launches the weight-gradient GEMM and returns the f32 panel.
    Returns the gradient tensor for autograd, or ``None`` after accumulating into ``weight.grad`` on the side stream."""
    g = weight.grad
    if not (ASYNC_WGRAD and g is not None and g.dtype == F32 and g.is_contiguous() and g.shape == weight.shape):
        return unpack_wgrad(desc, run_gemm(), weight)
    dev = weight.device
    main, side = torch.cuda.current_stream(dev), side_stream(dev)
    side.wait_stream(main)                      # dz / activations produced so far are visible to the side stream
    global _WGRAD_OVERLAPPED
    with torch.cuda.stream(side):
        _WGRAD_OVERLAPPED = True
        try:
            dwp = run_gemm()
        finally:
            _WGRAD_OVERLAPPED = False
        ns, st = _slabs_of(dwp)
        L.check(L.lib.uclstm_unpack_wgrad(C.byref(desc), _p(dwp), ns, st, _p(g), 1, _stream()), "unpack_wgrad(accumulate)")
        for t in inputs:
            if t is not None:
                t.record_stream(side)           # the caching allocator must not recycle them under the side stream
        for hook in GRAD_SIDE_HOOKS:
            hook(weight)
    _schedule_join(dev)
    return None


def direct_grad(param: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    """``param.grad`` when a backward kernel may accumulate into it directly (pre-attached contiguous f32 buffer, as
    ``optim.FlatParams`` provides), else ``None``.  Saves autograd's tiny per-parameter accumulate kernels."""
    if param is None or not DIRECT_GRADS:
        return None
    g = param.grad
    if g is None or g.dtype != F32 or not g.is_contiguous() or g.shape != param.shape:
        return None
    return g


def grad_written(param: torch.Tensor) -> None:
    """Tell data-parallel code that ``param.grad`` has been written outside autograd's accumulator."""
    for hook in GRAD_SIDE_HOOKS:
        hook(param)


# A parameter whose gradient is written outside autograd announces itself once per USE (autograd's own accumulator fires
# once per backward pass, however often the parameter was used).  Operators therefore report every use in forward, so that
# a data-parallel wrapper knows how many announcements complete a parameter -- a module called twice before one backward
# (reference-style per-timestep loops over ConvLSTMCell, two forwards under one loss) must not release its bucket early.
USE_HOOKS: list = []


def note_use(*params) -> None:
    if USE_HOOKS:
        for p in params:
            if p is not None:
                for hook in USE_HOOKS:
                    hook(p)


# ---------------------------------------------------------------------------------------------
# raw launches
# ---------------------------------------------------------------------------------------------
# A single GEMM launch addresses every operand through a 32-bit buffer descriptor whose bit 31 means "outside the tensor"
# (that is how padding is free), so one launch sees at most ~2 GiB of each tensor.  The reference, which runs one timestep
# at a time, has no such limit: batched launches over n_img = T*B images are therefore cut into image ranges here -- on
# BatchNorm-group (timestep) boundaries when the launch also produces statistics, whose rows are per group anyway.
LAUNCH_BYTES_LIMIT = (1 << 31) - (1 << 23)


def _img_chunks(n_img: int, groups: int, bytes_per_img: int, what: str, whole_groups: bool = False) -> List[Tuple[int, int]]:
    """Image ranges [i0, i1) of at most LAUNCH_BYTES_LIMIT bytes each; with ``whole_groups`` (the launch writes per-group
    BatchNorm partial sums) a range is a whole number of groups of n_img/groups images."""
    if n_img * bytes_per_img < LAUNCH_BYTES_LIMIT:
        return [(0, n_img)]
    ipg = n_img // groups
    unit = ipg if whole_groups else 1
    per = ((LAUNCH_BYTES_LIMIT - 1) // (bytes_per_img * unit)) * unit
    if per < unit or per == 0:
        raise L.UclstmError(f"{what}: one {'BatchNorm group' if whole_groups else 'image'} of this tensor ({unit} image(s) x "
                            f"{bytes_per_img} bytes) exceeds the {LAUNCH_BYTES_LIMIT}-byte range a single launch can address; "
                            "use a smaller per-GPU batch")
    return [(i, min(n_img, i + per)) for i in range(0, n_img, per)]


def _bytes_per_img(t: torch.Tensor) -> int:
    return t[0].numel() * t.element_size()


def _same_act_dtype(tensors, what: str) -> None:
    dt = tensors[0].dtype
    if dt not in ACT or any(t.dtype != dt for t in tensors):
        raise L.UclstmError(f"{what}: activations, panels and outputs must share one 16-bit dtype, got {[str(t.dtype) for t in tensors]}")


# Split-K for STORE-epilogue convolutions whose tile grid cannot fill the chip (no BatchNorm statistics wanted: inference, or
# evaluation-mode statistics): below SPLITK_STORE_BELOW 128 x 128 tiles.  UCLSTM_SPLITK_STORE=0 switches it off.
SPLITK_STORE = os.environ.get("UCLSTM_SPLITK_STORE", "1") != "0"
SPLITK_STORE_BELOW = int(os.environ.get("UCLSTM_SPLITK_STORE_BELOW", "192"))
# Second rule, OFF by default (UCLSTM_SPLITK_STORE_LONGK=288 switches it on): a grid of 128 x 256 tiles that needs a second round of the
# 256 CUs for at most half a round's worth of tiles, on a LONG K, as two K ranges.  Measured (tools/bench_store_splitk.py,
# profiles/round3_store_splitk.txt, finish pass included): alone, 320 tiles x K = 36864 (the temporal ConvLSTM's input gradient over all
# timesteps) 823 -> 632 us and 384 tiles (the same layer at 256 x 256, B = 4) 792 -> 741 us; 320 tiles x K = 9216 +-3 %; every other
# under-filled shape (160 / 320 / 640 tiles, K <= 18432) LOSES 2 - 70 %.  In the STEP, where the weight-gradient stream runs beside
# the main stream, the rule fires (igemm_fwd_store_splitk, 0.786 -> 0.611 ms serialised) and the backward phase gets no shorter:
# 20.643 -> 20.678 ms, slower in three of three same-box pairs (profiles/round3_store_splitk_step_ab.txt) -- the CUs the under-filled
# round leaves idle are taken by the other stream, and the slabs + finish pass are extra work.  Useful only for serialised /
# single-stream execution, hence a switch and not the default.
SPLITK_STORE_LONGK_STEPS = int(os.environ.get("UCLSTM_SPLITK_STORE_LONGK", "0"))          # K-steps of 64 from which it applies; 0 = off


def store_split_k(pixels: int, N: int, ksteps: int) -> int:
    """K ranges of a store-epilogue convolution without BatchNorm statistics (1 = one pass).  Pure function of the shape."""
    ks = split_k_factor(pixels, N, ksteps, min_blocks=SPLITK_STORE_BELOW)
    if ks == 1 and SPLITK_STORE_LONGK_STEPS > 0 and ksteps >= SPLITK_STORE_LONGK_STEPS:
        tiles = ((pixels + 127) // 128) * ((N + 255) // 256)
        if 256 < tiles <= 384:
            ks = 2
    return ks


def igemm_store(srcs: Sequence[SrcView], wp: torch.Tensor, out_hw: Tuple[int, int], n_img: int, segs, *, ktap: int, scale: int = 1,
                pad: int = 0, groups: int = 1, bias=None, col_scale=None, col_shift=None, relu: bool = False, stats=None) -> None:
    """segs: list of (tensor, n_begin, n_end, c_off, scale, oy, ox).  ``stats``: [groups, tiles_per_group, N, 2]."""
    per_img = max([_bytes_per_img(sv.t) for sv in srcs] + [_bytes_per_img(sg[0]) for sg in segs])
    chunks = _img_chunks(n_img, groups, per_img, "igemm_fwd(store)", whole_groups=stats is not None)
    if len(chunks) > 1:
        if LAUNCH_LOG is not None:
            LAUNCH_LOG.append(("split", "igemm_fwd(store)", len(chunks)))
        ipg = n_img // groups
        for i0, i1 in chunks:
            sub_src = [SrcView(sv.t[i0:i1], sv.offY, sv.offX) for sv in srcs]
            sub_seg = [(sg[0][i0:i1],) + tuple(sg[1:]) for sg in segs]
            sub_stats = None if stats is None else stats[i0 // ipg:i1 // ipg]
            igemm_store(sub_src, wp, out_hw, i1 - i0, sub_seg, ktap=ktap, scale=scale, pad=pad,
                        groups=(i1 - i0) // ipg if stats is not None else 1, bias=bias, col_scale=col_scale, col_shift=col_shift,
                        relu=relu, stats=sub_stats)
        return
    if stats is None and SPLITK_STORE and scale == 1 and len(segs) == 1:
        # few output pixels (batch-1 inference: a 32 x 32 bottleneck is 4 x 8 tiles for 256 CUs): K ranges store f32 partial
        # tiles, uclstm_splitk_finish applies the epilogue.  Only for one dense destination (a plain convolution).
        t, n_begin, n_end, c_off, sg_scale, oy, ox = (tuple(segs[0]) + (0, 1, 0, 0))[:7]
        N, pixels = wp.shape[0], n_img * out_hw[0] * out_hw[1]
        if (n_begin, n_end, c_off, sg_scale, oy, ox) == (0, N, 0, 1, 0, 0) and t.is_contiguous() and tuple(t.shape) == (n_img, out_hw[0], out_hw[1], N):
            ksplit = store_split_k(pixels, N, wp.shape[1] // 64)
            if ksplit > 1:
                nsl = ksplit_used(wp.shape[1], ksplit, ktap)
                pre = torch.empty((nsl, pixels, N), dtype=F32, device=t.device)
                igemm_atomic(srcs, wp, out_hw, n_img, pre, ksplit, ktap=ktap, pad=pad, slabs=True, kind="igemm_fwd_store_splitk")
                L.check(_k(wp).uclstm_splitk_finish(_p(pre), nsl, pre.stride(0), N, _p(bias), _p(col_scale), _p(col_shift), int(relu), _p(t),
                                                    pixels, N, _stream()), "splitk_finish")
                return
    d = L.IgemmDesc()
    d.n_img, d.H, d.W, d.groups = n_img, out_hw[0], out_hw[1], groups
    d.ktap, d.scale, d.pad, d.nsrc = ktap, scale, pad, len(srcs)
    for i, s in enumerate(srcs):
        s.fill(d.src[i])
    d.wp, d.N, d.Ktot = wp.data_ptr(), wp.shape[0], wp.shape[1]
    d.bias = None if bias is None else bias.data_ptr()
    d.col_scale = None if col_scale is None else col_scale.data_ptr()
    d.col_shift = None if col_shift is None else col_shift.data_ptr()
    d.relu, d.epi = int(relu), L.EPI_STORE
    d.nseg = len(segs)
    for i, sg in enumerate(segs):
        _fill_seg(d.seg[i], *sg)
    d.stats = None if stats is None else stats.data_ptr()
    flops = 2.0 * n_img * out_hw[0] * out_hw[1] * d.N * ktap * ktap * sum(s.t.shape[3] for s in srcs)
    _log_shape(d)
    _same_act_dtype([sv.t for sv in srcs] + [wp] + [sg[0] for sg in segs], "igemm_fwd(store)")
    K = _k(wp)
    _timed(_kernel_kind("igemm_fwd_store", d), flops, lambda: L.check(K.uclstm_igemm_fwd(C.byref(d), _stream()), "igemm_fwd(store)"),
           f"M={n_img * out_hw[0] * out_hw[1]} N={d.N} K={d.Ktot} ktap={ktap} nsrc={len(srcs)} nseg={len(segs)}",
           nbytes=_nb(*[sv.t for sv in srcs], wp) + 2.0 * n_img * out_hw[0] * out_hw[1] * d.N if PROFILE is not None else 0.0)


def split_k_factor(pixels: int, N: int, ksteps: int, min_blocks: int = 384, target: int = 512) -> int:
    """K ranges for a GEMM whose 128x128 tile grid alone cannot fill the 256 CUs (the per-timestep GEMMs of the
    ConvLSTM recurrence: M = B*h*w is as small as 512).  1 = do not split."""
    tiles = ((pixels + 127) // 128) * ((N + 127) // 128)
    if tiles >= min_blocks:
        return 1
    return max(1, min((target + tiles - 1) // tiles, ksteps // 8))


def ksplit_used(Ktot: int, ksplit: int, ktap: int = 3) -> int:
    """K ranges the library really uses for a requested split (whole chunks of ktap^2 K-steps, every range non-empty) = slab
    count of ``slabs=True``."""
    n = int(L.lib.uclstm_igemm_ksplit_used(Ktot, ktap, ksplit))
    if n < 1:
        raise L.UclstmError(f"igemm_ksplit_used({Ktot}, {ktap}, {ksplit}): bad argument")
    return n


def _atomic_desc(srcs: Sequence[SrcView], wp: torch.Tensor, out_hw: Tuple[int, int], n_img: int, acc_out: torch.Tensor, ksplit: int, *,
                 ktap: int, scale: int = 1, pad: int = 0, slabs: bool = False):
    """Descriptor of a split-K launch: (desc, algorithmic flops, algorithmic bytes, note, tensors to keep alive)."""
    _dev(acc_out, F32, "acc_out")
    d = L.IgemmDesc()
    d.n_img, d.H, d.W, d.groups = n_img, out_hw[0], out_hw[1], 1
    d.ktap, d.scale, d.pad, d.nsrc = ktap, scale, pad, len(srcs)
    for i, s in enumerate(srcs):
        s.fill(d.src[i])
    d.wp, d.N, d.Ktot = wp.data_ptr(), wp.shape[0], wp.shape[1]
    d.relu, d.epi, d.nseg = 0, L.EPI_ATOMIC, 0
    d.acc_out, d.acc_ld, d.ksplit = acc_out.data_ptr(), acc_out.shape[-1], ksplit
    d.acc_slab = acc_out.stride(0) if slabs else 0
    if slabs and (acc_out.dim() != 3 or acc_out.shape[0] != ksplit_used(d.Ktot, ksplit, ktap) or not acc_out.is_contiguous()):
        raise L.UclstmError("igemm_atomic(slabs=True): acc_out must be a contiguous [ksplit_used, pixels, ld] tensor")
    flops = 2.0 * n_img * out_hw[0] * out_hw[1] * d.N * ktap * ktap * sum(s.t.shape[3] for s in srcs)
    _same_act_dtype([sv.t for sv in srcs] + [wp], "igemm_fwd(atomic)")
    nbytes = _nb(*[sv.t for sv in srcs], wp) + 4.0 * n_img * out_hw[0] * out_hw[1] * d.N if PROFILE is not None else 0.0
    return d, flops, nbytes, f"M={n_img * out_hw[0] * out_hw[1]} N={d.N} K={d.Ktot} ktap={ktap} ksplit={ksplit}"


def igemm_atomic(srcs: Sequence[SrcView], wp: torch.Tensor, out_hw: Tuple[int, int], n_img: int, acc_out: torch.Tensor, ksplit: int, *,
                 ktap: int, scale: int = 1, pad: int = 0, slabs: bool = False, kind: str = "igemm_fwd_atomic") -> None:
    """Split-K convolution as `ksplit` K ranges.  ``slabs=False``: acc_out[pixel, n] += ... with f32 atomics
    (acc_out f32 [pixels, ld>=N], zeroed by the caller).  ``slabs=True``: acc_out is [ksplit_used, pixels, ld]; range r
    stores into acc_out[r] and the consumer adds the slabs (plain stores run ~4.6x faster than float atomics)."""
    d, flops, nbytes, note = _atomic_desc(srcs, wp, out_hw, n_img, acc_out, ksplit, ktap=ktap, scale=scale, pad=pad, slabs=slabs)
    _log_shape(d)
    K = _k(wp)
    _timed(_kernel_kind(kind, d), flops, lambda: L.check(K.uclstm_igemm_fwd(C.byref(d), _stream()), "igemm_fwd(atomic)"), note, nbytes=nbytes)


def _lstm_desc(x: Optional[torch.Tensor], h_prev: torch.Tensor, wp: torch.Tensor, bias: Optional[torch.Tensor], c_prev: Optional[torch.Tensor],
               c_out: torch.Tensor, h_out: torch.Tensor, gates_out: Optional[torch.Tensor], ksize: int = 3,
               pre_add: Optional[torch.Tensor] = None):
    d = L.IgemmDesc()
    B, H, W, _ = h_prev.shape
    d.n_img, d.H, d.W, d.groups = B, H, W, 1
    d.ktap, d.scale, d.pad = ksize, 1, ksize // 2
    if x is not None:
        d.nsrc = 2
        SrcView(x).fill(d.src[0])
        SrcView(h_prev).fill(d.src[1])
    else:
        d.nsrc = 1
        SrcView(h_prev).fill(d.src[0])
    d.pre_add = None if pre_add is None else _dev(pre_add, F32, "pre_add").data_ptr()
    d.wp, d.N, d.Ktot = wp.data_ptr(), wp.shape[0], wp.shape[1]
    d.bias = None if bias is None else bias.data_ptr()
    d.relu, d.epi, d.nseg = 0, L.EPI_LSTM, 0
    d.Hd_p = h_prev.shape[3]
    d.c_prev = None if c_prev is None else c_prev.data_ptr()
    d.c_out, d.h_out = c_out.data_ptr(), h_out.data_ptr()
    d.gates_out = None if gates_out is None else gates_out.data_ptr()
    flops = 2.0 * B * H * W * (4 * d.Hd_p) * ksize * ksize * ((x.shape[3] if x is not None else 0) + h_prev.shape[3])
    _same_act_dtype([h_prev, wp, h_out] + ([x] if x is not None else []) + ([gates_out] if gates_out is not None else []), "igemm_fwd(lstm)")
    nbytes = _nb(x, h_prev, wp, c_prev, c_out, h_out, gates_out, pre_add) if PROFILE is not None else 0.0
    return d, flops, nbytes, f"M={B * H * W} N={d.N} K={d.Ktot}"


def igemm_lstm(x: Optional[torch.Tensor], h_prev: torch.Tensor, wp: torch.Tensor, bias: Optional[torch.Tensor], c_prev: Optional[torch.Tensor],
               c_out: torch.Tensor, h_out: torch.Tensor, gates_out: Optional[torch.Tensor], ksize: int = 3,
               pre_add: Optional[torch.Tensor] = None) -> None:
    """Fused ConvLSTM cell step.  ``x`` given: gate conv over (x_t, h_{t-1}) with the two-source panel.  ``x=None``: the
    launch carries W_h * h_{t-1} only and ``pre_add`` (f32 [pixels, N]) holds the hoisted W_x * x_t."""
    d, flops, nbytes, note = _lstm_desc(x, h_prev, wp, bias, c_prev, c_out, h_out, gates_out, ksize, pre_add)
    _log_shape(d)
    K = _k(wp)
    _timed(_kernel_kind("igemm_fwd_lstm", d), flops, lambda: L.check(K.uclstm_igemm_fwd(C.byref(d), _stream()), "igemm_fwd(lstm)"), note,
           nbytes=nbytes)


def igemm_group(items, K, kind: str = "igemm_fwd_group") -> None:
    """``items``: [(desc, flops, nbytes, note)] of INDEPENDENT launches -> one uclstm_igemm_fwd_group launch."""
    arr = (L.IgemmDesc * len(items))(*[it[0] for it in items])
    for it in items:
        _log_shape(it[0])
    _timed(kind + "[patch128x256]" if PROFILE is not None else kind, sum(it[1] for it in items),
           lambda: L.check(K.uclstm_igemm_fwd_group(arr, len(items), _stream()), "igemm_fwd_group"),
           " | ".join(it[3] for it in items), nbytes=sum(it[2] for it in items))


def group_launchable(descs) -> bool:
    """True when uclstm_igemm_fwd_group accepts these descriptors as one launch (all on the patch shape, same source count)."""
    arr = (L.IgemmDesc * len(descs))(*descs)
    return int(L.lib.uclstm_igemm_fwd_group_blocks(arr, len(descs))) > 0


def igemm_wgrad(srcs: Sequence[SrcView], dy_segs, N: int, Ktot: int, out_hw: Tuple[int, int], n_img: int, *, ktap: int, scale: int = 1,
                pad: int = 0) -> torch.Tensor:
    """Weight-gradient GEMM; returns the f32 panel gradient as pixel-range slabs [splits, N, Ktot] (the unpack adds them).
    Operands beyond the single-launch byte range are processed as image ranges, each range contributing its own slabs."""
    dev = srcs[0].t.device
    _same_act_dtype([sv.t for sv in srcs] + [sg[0] for sg in dy_segs], "igemm_wgrad")
    K = _k(srcs[0].t)
    per_img = max([_bytes_per_img(sv.t) for sv in srcs] + [_bytes_per_img(sg[0]) for sg in dy_segs])
    chunks = _img_chunks(n_img, 1, per_img, "igemm_wgrad")
    descs = []
    for i0, i1 in chunks:
        d = L.WgradDesc()
        d.n_img, d.H, d.W = i1 - i0, out_hw[0], out_hw[1]
        d.ktap, d.scale, d.pad, d.nsrc = ktap, scale, pad, len(srcs)
        for i, sv in enumerate(srcs):
            (sv if len(chunks) == 1 else SrcView(sv.t[i0:i1], sv.offY, sv.offX)).fill(d.src[i])
        d.N, d.Ktot = N, Ktot
        d.nseg = len(dy_segs)
        for i, sg in enumerate(dy_segs):
            _fill_seg(d.seg[i], *(sg if len(chunks) == 1 else (sg[0][i0:i1],) + tuple(sg[1:])))
        # slab mode: every pixel range stores its partial panel into its own slab (no float atomics, nothing to zero);
        # the library picks the range count for its tile shape and the 256 CUs, uclstm_unpack_wgrad adds the slabs
        d.splits, d.accumulate, d.slab = 0, 1, N * Ktot
        d.overlapped = int(_WGRAD_OVERLAPPED)          # on the side stream: the plan that interferes least with the main stream
        splits = int(L.lib.uclstm_igemm_wgrad_splits(C.byref(d)))
        if splits < 1:
            raise L.UclstmError(f"igemm_wgrad: bad descriptor (code {splits})")
        d.splits = splits
        descs.append(d)
    if LAUNCH_LOG is not None:
        if len(chunks) > 1:
            LAUNCH_LOG.append(("split", "igemm_wgrad", len(chunks)))
        for d in descs:
            LAUNCH_LOG.append(("wgrad", int(L.lib.uclstm_igemm_wgrad_shape(C.byref(d))), N, Ktot, int(d.splits)))
    total = sum(d.splits for d in descs)
    dwp = torch.empty((total, N, Ktot), dtype=F32, device=dev)
    taps = ktap * ktap
    at = 0
    for d in descs:
        d.dwp = dwp[at].data_ptr()
        at += d.splits
        flops = 2.0 * d.n_img * out_hw[0] * out_hw[1] * N * taps * sum(s.t.shape[3] for s in srcs)
        kind = "igemm_wgrad"
        if PROFILE is not None:          # rocprofv3 names: igemm_wgrad_p3_kernel / igemm_wgrad_p2_kernel<1 or 2, nsrc> / igemm_wgrad_kernel
            kind += "[" + {4: "ring64", 3: "p3_256x256", 2: "p2_128x128", 1: "p2_64x256", 0: "generic"}.get(int(L.lib.uclstm_igemm_wgrad_shape(C.byref(d))), "?") + "]"
        _timed(kind, flops, lambda d=d: L.check(K.uclstm_igemm_wgrad(C.byref(d), _stream()), "igemm_wgrad"),
               f"M={d.n_img * out_hw[0] * out_hw[1]} N={N} K={Ktot} ktap={ktap} splits={d.splits}",
               nbytes=(_nb(*[sv.t for sv in srcs], *[sg[0] for sg in dy_segs]) * (d.n_img / n_img) + 4.0 * N * Ktot) if PROFILE is not None else 0.0)
    return dwp


def colsum(a: torch.Tensor, into: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Column sums of a 16-bit [pixels, Cp] tensor.  ``into``: f32 [Cp] buffer the sums are ADDED to (e.g. an attached bias
    gradient: saves the zero-fill, the slice copy and autograd's accumulate kernel); else a fresh zeroed vector."""
    _dev(a, ACT, "colsum input")
    Cp = a.shape[-1]
    out = into if into is not None else torch.zeros((Cp,), dtype=F32, device=a.device)
    L.check(_k(a).uclstm_colsum(_p(a), _p(out), a.numel() // Cp, Cp, _stream()), "colsum")
    return out


def bias_grad_from_colsum(a: torch.Tensor, bias: Optional[torch.Tensor], valid: int) -> Optional[torch.Tensor]:
    """Bias gradient = column sums of ``a`` ([..., Cp], ``valid`` real channels).  When the bias owns an attached f32 gradient
    of exactly Cp elements the kernel accumulates straight into it and ``None`` is returned for autograd."""
    if bias is None:
        return None
    g = direct_grad(bias)
    if g is not None and g.numel() == a.shape[-1] == valid:
        # (moving this HBM-bound pass to the weight-gradient stream was measured: no gain, unlike the tiny BatchNorm
        # parameter-gradient kernels below)
        colsum(a, into=g.view(-1))
        grad_written(bias)
        return None
    return colsum(a)[:valid].contiguous()


# ---------------------------------------------------------------------------------------------
# layout conversion at the module boundary
# ---------------------------------------------------------------------------------------------
class ToNHWC(torch.autograd.Function):
    """f32 NCHW [N,C,H,W] -> bf16 NHWC [N,H,W,Cp]."""

    @staticmethod
    def forward(ctx, x):
        _dev(x, F32, "input")
        N, Cc, H, W = x.shape
        out = torch.empty((N, H, W, cpad(Cc)), dtype=_COMPUTE_DTYPE, device=x.device)
        L.check(_k(out).uclstm_nchw_to_nhwc(_p(x), _p(out), N, Cc, cpad(Cc), H, W, N, 0, Cc * H * W, _stream()), "nchw_to_nhwc")
        ctx.C = Cc
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        N, H, W, Cp = g.shape
        out = torch.empty((N, ctx.C, H, W), dtype=F32, device=g.device)
        L.check(_k(g).uclstm_nhwc_to_nchw(_p(g), _p(out), N, ctx.C, Cp, H, W, _stream()), "nhwc_to_nchw")
        return out


class FromNHWC(torch.autograd.Function):
    """bf16 NHWC [N,H,W,Cp] -> f32 NCHW [N,C,H,W]."""

    @staticmethod
    def forward(ctx, a, Cc):
        _dev(a, ACT, "activation")
        N, H, W, Cp = a.shape
        out = torch.empty((N, Cc, H, W), dtype=F32, device=a.device)
        L.check(_k(a).uclstm_nhwc_to_nchw(_p(a), _p(out), N, Cc, Cp, H, W, _stream()), "nhwc_to_nchw")
        ctx.Cp = Cp
        ctx.act_dtype = a.dtype
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        N, Cc, H, W = g.shape
        out = torch.empty((N, H, W, ctx.Cp), dtype=ctx.act_dtype, device=g.device)
        L.check(_k(out).uclstm_nchw_grad_to_nhwc(_p(g), _p(out), N, Cc, ctx.Cp, H, W, _stream()), "nchw_grad_to_nhwc")
        return out, None


class StateToNHWC(torch.autograd.Function):
    """f32 NCHW cell state -> f32 NHWC [N,H,W,Cp]."""

    @staticmethod
    def forward(ctx, c):
        _dev(c, F32, "cell state")
        N, Cc, H, W = c.shape
        out = torch.empty((N, H, W, cpad(Cc)), dtype=F32, device=c.device)
        L.check(L.lib.uclstm_nchw_to_nhwc_f32(_p(c), _p(out), N, Cc, cpad(Cc), H, W, _stream()), "nchw_to_nhwc_f32")
        ctx.C = Cc
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        N, H, W, Cp = g.shape
        out = torch.empty((N, ctx.C, H, W), dtype=F32, device=g.device)
        L.check(L.lib.uclstm_nhwc_to_nchw_f32(_p(g), _p(out), N, ctx.C, Cp, H, W, _stream()), "nhwc_to_nchw_f32")
        return out


class StateFromNHWC(torch.autograd.Function):
    @staticmethod
    def forward(ctx, c, Cc):
        _dev(c, F32, "cell state")
        N, H, W, Cp = c.shape
        out = torch.empty((N, Cc, H, W), dtype=F32, device=c.device)
        L.check(L.lib.uclstm_nhwc_to_nchw_f32(_p(c), _p(out), N, Cc, Cp, H, W, _stream()), "nhwc_to_nchw_f32")
        ctx.Cp = Cp
        return out

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        N, Cc, H, W = g.shape
        out = torch.empty((N, H, W, ctx.Cp), dtype=F32, device=g.device)
        L.check(L.lib.uclstm_nchw_to_nhwc_f32(_p(g), _p(out), N, Cc, ctx.Cp, H, W, _stream()), "nchw_to_nhwc_f32")
        return out, None


def im2col_first(x: torch.Tensor, time_major: bool) -> torch.Tensor:
    """First-layer gather of a f32 input that needs no gradient.

    ``x`` is ``[N,C,H,W]`` or, with ``time_major``, ``[B,T,C,H,W]`` read as image ``t*B+b``
    (train/unet.py:180 feeds ``x_seq[:, t]``).  Returns bf16 ``[N,H,W,Kp]`` with K = (tap, c).
    """
    _dev(x, F32, "input")
    if time_major:
        B, T, Cc, H, W = x.shape
        n_img, inner, inner_stride, outer_stride = B * T, B, Cc * H * W, T * Cc * H * W
    else:
        n_img, Cc, H, W = x.shape
        inner, inner_stride, outer_stride = n_img, 0, Cc * H * W
    Kp = cpad(9 * Cc)
    out = torch.empty((n_img, H, W, Kp), dtype=_COMPUTE_DTYPE, device=x.device)
    L.check(_k(out).uclstm_im2col3x3_first(_p(x), _p(out), n_img, Cc, Kp, H, W, inner, inner_stride, outer_stride, _stream()),
            "im2col3x3_first")
    return out


# ---------------------------------------------------------------------------------------------
# conv3x3 (+ optional second source) + BatchNorm + ReLU   (train/unet.py:70-71, :98)
# ---------------------------------------------------------------------------------------------
def _eval_bn_constants(gamma, beta, running_mean, running_var, eps, momentum, Co, Cop) -> torch.Tensor:
    """[2, 1, Cop] f32 (scale, shift) of an evaluation-mode BatchNorm, folded into the conv epilogue.  Constants of frozen
    weights: kept in the current PanelCache (StreamingPredictor) like the packed panels, so a rollout frame does not rebuild them."""
    cache, key, ver = _ACTIVE_CACHE, None, None
    if cache is not None and not cache.stale():
        key = ("bn_eval", gamma.data_ptr(), beta.data_ptr(), running_mean.data_ptr(), running_var.data_ptr(), float(eps), Cop)
        ver = (gamma._version, beta._version, running_mean._version, running_var._version)
        hit = cache.entries.get(key)
        if hit is not None and hit[0] == ver:
            return hit[1]
    par = torch.empty((2, 1, Cop), dtype=F32, device=gamma.device)
    L.check(L.lib.uclstm_bn_finalize(None, 1, 0, Cop, Co, 0, _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                                     momentum, eps, _p(par[0]), _p(par[1]), None, None, _stream()), "bn_finalize(eval)")
    if key is not None:
        cache.entries[key] = (ver, par)
    return par


_OUTER_GRAD = True


class _GradAwareFunction(torch.autograd.Function):
    """Function.forward runs with grad mode off and ctx.needs_input_grad ignores torch.no_grad(): remember the CALLER's grad
    mode, so that a forward under no_grad (validation, StreamingPredictor) takes the inference kernels, saves nothing and does
    not count as a use for the data-parallel bucket bookkeeping."""

    @classmethod
    def apply(cls, *args, **kwargs):
        global _OUTER_GRAD
        prev, _OUTER_GRAD = _OUTER_GRAD, torch.is_grad_enabled()
        try:
            return super(_GradAwareFunction, cls).apply(*args, **kwargs)
        finally:
            _OUTER_GRAD = prev


def _will_backward(ctx) -> bool:
    return _OUTER_GRAD and any(ctx.needs_input_grad)


class ConvBNReLU(_GradAwareFunction):
    """One (conv3x3 pad 1 + bias -> BatchNorm2d -> ReLU) stage on NHWC bf16.

    ``x1`` (optional) is channel-concatenated after ``x0`` and may be smaller, centred by
    ``(offY, offX)`` (the F.pad of train/unet.py:95-97).  ``groups`` = number of BatchNorm
    statistic groups along the image axis (timesteps).  ``im2col=True`` means ``x0`` is the
    pre-gathered first-layer tensor (K = 9*Cin) and the conv runs as a plain GEMM.
    """

    @staticmethod
    def forward(ctx, x0, x1, weight, bias, gamma, beta, running_mean, running_var, c_valid, off, groups, training, momentum, eps,
                im2col, head_w=None, head_b=None, pool=False):
        # pool: also return MaxPool2d(2) of the activation -- (a, p) -- with the pooling fused into the BatchNorm kernels both ways
        # (uclstm_bn_apply_relu_pool / uclstm_bn_pool_bwd_*); every encoder block output feeds the next block's pooling and a skip.
        # head_w / head_b: the model's 1x1 output convolution (ONE output channel) fused into this stage (training mode only,
        # uclstm_bn_head_*): the stage then returns y f32 [n_img, 1, H, W] instead of its activation, which never exists.
        _dev(x0, ACT, "x0")
        Co, Ci_total = weight.shape[0], weight.shape[1]
        Cop = cpad(Co)
        n_img, H, W, _ = x0.shape
        dev = x0.device
        if im2col:
            pd = im2col_pack_desc(Co, Ci_total, x0.shape[3])
            srcs = [SrcView(x0)]
            ktap, pad = 1, 0
        else:
            c_pad = [x0.shape[3]] + ([x1.shape[3]] if x1 is not None else [])
            pd = conv_pack_desc(Co, Ci_total, list(c_valid), c_pad)
            srcs = [SrcView(x0)] + ([SrcView(x1, off[0], off[1])] if x1 is not None else [])
            ktap, pad = 3, 1
        wp = pack_weights(pd, weight, 0, x0.dtype)
        bp = pack_bias(pd, bias) if bias is not None else None
        out = torch.empty((n_img, H, W, Cop), dtype=x0.dtype, device=dev)
        K = _k(x0)
        need_bw = _will_backward(ctx)
        pooled = None
        if training:
            ppg = (n_img // groups) * H * W
            tpg = L.lib.uclstm_igemm_tiles_per_group(n_img, H, W, groups, Cop)
            stats = torch.empty((groups, tpg, Cop, 2), dtype=F32, device=dev)
            z = out
            igemm_store(srcs, wp, (H, W), n_img, [(z, 0, Cop, 0, 1, 0, 0)], ktap=ktap, pad=pad, groups=groups, bias=bp, stats=stats)
            par = torch.empty((4, groups, Cop), dtype=F32, device=dev)     # scale, shift, mean, rstd
            if ASYNC_WGRAD and BN_RUNNING_ON_SIDE and not torch.cuda.is_current_stream_capturing():
                # critical part in one launch (reduction + scale / shift / mean / rstd); the in-order running-statistics recursion,
                # which nothing in this step waits for, on the second stream (joined at the end of the forward pass: join_forward_side)
                L.check(L.lib.uclstm_bn_stats_fwd(_p(stats), groups, tpg, Cop, Co, ppg, _p(gamma), _p(beta), eps, _p(par[0]), _p(par[1]),
                                                  _p(par[2]), _p(par[3]), _stream()), "bn_stats_fwd")
                main, side = torch.cuda.current_stream(dev), side_stream(dev)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    L.check(L.lib.uclstm_bn_running_stats(_p(stats), groups, tpg, Cop, Co, ppg, _p(running_mean), _p(running_var), momentum,
                                                          _stream()), "bn_running_stats")
                    stats.record_stream(side)
                _FWD_SIDE_PENDING.add(str(dev))
            else:
                L.check(L.lib.uclstm_bn_finalize(_p(stats), groups, tpg, Cop, Co, ppg, _p(gamma), _p(beta), _p(running_mean),
                                                 _p(running_var), momentum, eps, _p(par[0]), _p(par[1]), _p(par[2]), _p(par[3]),
                                                 _stream()), "bn_finalize")
            if head_w is not None:
                a = torch.empty((n_img, 1, H, W), dtype=F32, device=dev)
                _timed_hbm("bn_head_fwd", 2.0 * z.numel() + 4.0 * n_img * H * W,
                           lambda: L.check(K.uclstm_bn_head_fwd(_p(z), _p(par[0]), _p(par[1]), _p(head_w), _p(head_b), _p(a), n_img * H * W, ppg,
                                                                Cop, Co, _stream()), "bn_head_fwd"))
            elif pool and H % 2 == 0 and W % 2 == 0 and Cop // 8 <= 256:
                a = torch.empty_like(z)
                pooled = torch.empty((n_img, H // 2, W // 2, Cop), dtype=z.dtype, device=dev)
                _timed_hbm("bn_apply_relu_pool", 4.5 * z.numel(),
                           lambda: L.check(K.uclstm_bn_apply_relu_pool(_p(z), _p(a), _p(pooled), _p(par[0]), _p(par[1]), n_img, H, W, Cop,
                                                                       groups, _stream()), "bn_apply_relu_pool"))
            else:
                a = torch.empty_like(z)
                _timed_hbm("bn_apply_relu", 4.0 * z.numel(),        # 2 B read + 2 B written per element
                           lambda: L.check(K.uclstm_bn_apply_relu(_p(z), _p(a), _p(par[0]), _p(par[1]), n_img * H * W, ppg, Cop, _stream()),
                                           "bn_apply_relu"))
            ctx.save_for_backward(x0, x1, weight, z, par, gamma, beta, bias, head_w, head_b)
            if need_bw:
                note_use(weight, gamma, beta, bias, head_w, head_b)
        elif head_w is not None:
            raise L.UclstmError("ConvBNReLU: the fused output head is a training-mode path")
        elif need_bw:
            join_forward_side(dev)
            # evaluation-mode statistics WITH a backward pass (fine-tuning through frozen BatchNorm): keep the pre-BN conv
            # output like the training path does, normalise with the running statistics as one group
            groups = 1
            z = out
            igemm_store(srcs, wp, (H, W), n_img, [(z, 0, Cop, 0, 1, 0, 0)], ktap=ktap, pad=pad, groups=1, bias=bp)
            par = torch.empty((4, 1, Cop), dtype=F32, device=dev)
            L.check(L.lib.uclstm_bn_finalize(None, 1, 0, Cop, Co, 0, _p(gamma), _p(beta), _p(running_mean), _p(running_var),
                                             momentum, eps, _p(par[0]), _p(par[1]), _p(par[2]), _p(par[3]), _stream()), "bn_finalize(eval)")
            a = torch.empty_like(z)
            L.check(K.uclstm_bn_apply_relu(_p(z), _p(a), _p(par[0]), _p(par[1]), n_img * H * W, n_img * H * W, Cop, _stream()),
                    "bn_apply_relu")
            ctx.save_for_backward(x0, x1, weight, z, par, gamma, beta, bias, None, None)
            note_use(weight, gamma, beta, bias)
        else:
            join_forward_side(dev)
            par = _eval_bn_constants(gamma, beta, running_mean, running_var, eps, momentum, Co, Cop)
            a = out
            igemm_store(srcs, wp, (H, W), n_img, [(a, 0, Cop, 0, 1, 0, 0)], ktap=ktap, pad=pad, groups=1, bias=bp,
                        col_scale=par[0], col_shift=par[1], relu=True)
            ctx.save_for_backward(x0, x1, weight, None, None, gamma, beta, bias, None, None)
        ctx.cfg = (tuple(c_valid), tuple(off), groups, training, im2col, Co, Ci_total, bias is not None)
        ctx.pool = bool(pool)
        if pool:
            ctx.pool_fused = pooled is not None
            if pooled is None:          # evaluation mode, odd sizes: the stand-alone pooling kernel on the finished activation
                pooled = torch.empty((n_img, H // 2, W // 2, Cop), dtype=a.dtype, device=dev)
                L.check(K.uclstm_maxpool2_fwd(_p(a), _p(pooled), n_img, H, W, Cop, _stream()), "maxpool2_fwd")
                if need_bw:
                    ctx.pool_act = a
            return a, pooled
        return a

    @staticmethod
    def backward(ctx, da, dp=None):
        x0, x1, weight, z, par, gamma, beta, bias, head_w, head_b = ctx.saved_tensors
        c_valid, off, groups, training, im2col, Co, Ci_total, has_bias = ctx.cfg
        n_img, H, W, Cop = z.shape
        dev = z.device
        pixels, ppg = n_img * H * W, (n_img // groups) * H * W
        sums = torch.empty((groups, Cop, 2), dtype=F32, device=dev)
        partials = torch.empty((int(L.lib.uclstm_bn_bwd_reduce_rows(pixels, ppg)), Cop, 2), dtype=F32, device=dev)
        K = _k(z)
        dz = torch.empty_like(z)
        d_head_w = d_head_b = None
        if head_w is not None:
            # fused output head: `da` is dy f32 [n_img, 1, H, W]; the activation gradient bf16(dy * w) is formed on the fly
            dy = _dev(da.contiguous().float(), F32, "dy")
            gw, gb = direct_grad(head_w), (direct_grad(head_b) if head_b is not None else None)
            direct = gw is not None and gw.is_contiguous() and (head_b is None or gb is not None)
            dwh = gw if direct else torch.zeros_like(head_w, memory_format=torch.contiguous_format)
            dbh = gb if (direct and head_b is not None) else torch.zeros((1,), dtype=F32, device=dev)
            _timed_hbm("bn_head_bwd_reduce", 2.0 * z.numel() + 4.0 * pixels,
                       lambda: L.check(K.uclstm_bn_head_bwd_reduce(_p(z), _p(dy), _p(par[0]), _p(par[1]), _p(par[2]), _p(par[3]), _p(head_w),
                                                                   _p(partials), _p(sums), _p(dwh), _p(dbh), pixels, ppg, Cop, Co, _stream()),
                                       "bn_head_bwd_reduce"))
            _timed_hbm("bn_head_bwd_apply", 4.0 * z.numel() + 4.0 * pixels,
                       lambda: L.check(K.uclstm_bn_head_bwd_apply(_p(z), _p(dy), _p(par[0]), _p(par[1]), _p(par[2]), _p(par[3]), _p(sums),
                                                                  _p(head_w), _p(dz), pixels, ppg, Cop, Co, _stream()), "bn_head_bwd_apply"))
            if direct:
                grad_written(head_w)
                if head_b is not None:
                    grad_written(head_b)
            else:
                d_head_w, d_head_b = dwh, (dbh if head_b is not None else None)
        elif ctx.pool and dp is not None and ctx.pool_fused:
            # pooling fused: the gradient of the activation is (skip gradient) + scatter(dp), formed inside the BatchNorm kernels
            dsk = None if da is None else da.contiguous()
            dp = dp.contiguous()
            prt = torch.empty((int(L.lib.uclstm_bn_pool_bwd_rows(n_img, H, W, Cop, groups)), Cop, 2), dtype=F32, device=dev)
            _timed_hbm("bn_pool_bwd_reduce", (4.5 if dsk is not None else 2.5) * z.numel(),
                       lambda: L.check(K.uclstm_bn_pool_bwd_reduce(_p(z), _p(dsk), _p(dp), _p(par[0]), _p(par[1]), _p(par[2]), _p(par[3]),
                                                                   _p(prt), _p(sums), n_img, H, W, Cop, groups, _stream()),
                                       "bn_pool_bwd_reduce"))
            sums_dz = sums if training else torch.zeros_like(sums)
            _timed_hbm("bn_pool_bwd_apply", (6.5 if dsk is not None else 4.5) * z.numel(),
                       lambda: L.check(K.uclstm_bn_pool_bwd_apply(_p(z), _p(dsk), _p(dp), _p(par[0]), _p(par[1]), _p(par[2]), _p(par[3]),
                                                                  _p(sums_dz), _p(dz), n_img, H, W, Cop, groups, _stream()),
                                       "bn_pool_bwd_apply"))
        else:
            if ctx.pool and dp is not None:
                # pooling not fused (odd sizes / evaluation-mode statistics path): the stand-alone max-pool backward kernel first
                act = ctx.pool_act
                odd = bool(H % 2 or W % 2)
                dfull = torch.zeros_like(act) if odd else torch.empty_like(act)
                add = da.contiguous() if (da is not None and not odd) else None
                L.check(K.uclstm_maxpool2_bwd(_p(act), _p(dp.contiguous()), _p(add), _p(dfull), n_img, H, W, Cop, _stream()), "maxpool2_bwd")
                da = dfull if (da is None or add is not None) else dfull + da
            da = da.contiguous()
            _timed_hbm("bn_bwd_reduce", 4.0 * z.numel(),            # z and da read once
                       lambda: L.check(K.uclstm_bn_bwd_reduce(_p(z), _p(da), _p(par[0]), _p(par[1]), _p(par[2]), _p(par[3]), _p(partials),
                                                              _p(sums), pixels, ppg, Cop, _stream()), "bn_bwd_reduce"))
            # training: dz = scale*(g - s1/n - xhat*s2/n).  Evaluation-mode statistics are constants, the two mean terms vanish:
            # the same kernel with zero sums gives dz = scale*g (sums itself still holds dbeta / dgamma)
            sums_dz = sums if training else torch.zeros_like(sums)
            _timed_hbm("bn_bwd_apply", 6.0 * z.numel(),             # z, da read, dz written
                       lambda: L.check(K.uclstm_bn_bwd_apply(_p(z), _p(da), _p(par[0]), _p(par[1]), _p(par[2]), _p(par[3]), _p(sums_dz),
                                                             _p(dz), pixels, ppg, Cop, _stream()), "bn_bwd_apply"))
        # training: the conv bias feeds BatchNorm, which removes any per-channel constant -- its gradient is analytically 0.
        # With frozen statistics it is the column sum of dz.
        bias_grad = (lambda: colsum(dz)[:Co].contiguous()) if not training else (lambda: torch.zeros((Co,), dtype=F32, device=dev))
        g_gamma, g_beta = direct_grad(gamma), direct_grad(beta)
        if g_gamma is not None and g_beta is not None:
            # one kernel accumulates straight into the attached gradient buffers (instead of sum + 2 copies + 2 accumulates).
            # Nothing in the backward pass waits for it: with the weight-gradient stream in use it goes there (18 small
            # dependent launches off the main stream; joined with the weight gradients at the end of backward).
            if ASYNC_WGRAD and PARAM_GRADS_ON_SIDE:
                main, side = torch.cuda.current_stream(dev), side_stream(dev)
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    L.check(L.lib.uclstm_bn_bwd_param_grads(_p(sums), groups, Cop, Co, _p(g_gamma), _p(g_beta), 1, _stream()),
                            "bn_bwd_param_grads")
                    sums.record_stream(side)
                    grad_written(gamma)
                    grad_written(beta)
                _schedule_join(dev)
            else:
                L.check(L.lib.uclstm_bn_bwd_param_grads(_p(sums), groups, Cop, Co, _p(g_gamma), _p(g_beta), 1, _stream()), "bn_bwd_param_grads")
                grad_written(gamma)
                grad_written(beta)
            dgamma = dbeta = None
            dbias = None
            if has_bias:
                g_bias = direct_grad(bias)
                if g_bias is not None:
                    if not training:
                        g_bias.add_(bias_grad())
                    grad_written(bias)                          # training: += 0
                else:
                    dbias = bias_grad()
        else:
            tot = sums.sum(dim=0)
            dbeta = tot[:Co, 0].contiguous()
            dgamma = tot[:Co, 1].contiguous()
            dbias = bias_grad() if has_bias else None

        dy_seg = [(dz, 0, Cop, 0, 1, 0, 0)]
        if im2col:
            pd = im2col_pack_desc(Co, Ci_total, x0.shape[3])
            dweight = wgrad_into_param(weight, pd, [x0, dz], lambda: igemm_wgrad([SrcView(x0)], dy_seg, pd.N, pd.Ktot, (H, W), n_img,
                                                                                 ktap=1, pad=0))
        else:
            c_pad = [x0.shape[3]] + ([x1.shape[3]] if x1 is not None else [])
            pd = conv_pack_desc(Co, Ci_total, list(c_valid), c_pad)
            srcs = [SrcView(x0)] + ([SrcView(x1, off[0], off[1])] if x1 is not None else [])
            dweight = wgrad_into_param(weight, pd, [x0, x1, dz], lambda: igemm_wgrad(srcs, dy_seg, pd.N, pd.Ktot, (H, W), n_img,
                                                                                     ktap=3, pad=1))

        dx0 = dx1 = None
        if ctx.needs_input_grad[0] and not im2col:
            dd = conv_dgrad_pack_desc(Co, Ci_total, c_valid[0])
            wd = pack_weights(dd, weight, 0, dz.dtype)
            dx0 = torch.empty_like(x0)
            igemm_store([SrcView(dz)], wd, (H, W), n_img, [(dx0, 0, dd.N, 0, 1, 0, 0)], ktap=3, pad=1)
        if x1 is not None and ctx.needs_input_grad[1]:
            dd = conv_dgrad_pack_desc(Co, Ci_total, c_valid[1])
            wd = pack_weights(dd, weight, c_valid[0] * 9, dz.dtype)
            dx1 = torch.empty_like(x1)
            igemm_store([SrcView(dz)], wd, (H, W), n_img, [(dx1, 0, dd.N, 0, 1, -off[0], -off[1])], ktap=3, pad=1)
        return dx0, dx1, dweight, dbias, dgamma, dbeta, None, None, None, None, None, None, None, None, None, d_head_w, d_head_b, None


# ---------------------------------------------------------------------------------------------
# MaxPool2d(2)  (train/unet.py:81)
# ---------------------------------------------------------------------------------------------
class MaxPool2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a):
        _dev(a, ACT, "activation")
        N, H, W, Cp = a.shape
        p = torch.empty((N, H // 2, W // 2, Cp), dtype=a.dtype, device=a.device)
        L.check(_k(a).uclstm_maxpool2_fwd(_p(a), _p(p), N, H, W, Cp, _stream()), "maxpool2_fwd")
        ctx.save_for_backward(a)
        return p

    @staticmethod
    def backward(ctx, dp):
        (a,) = ctx.saved_tensors
        dp = dp.contiguous()
        N, H, W, Cp = a.shape
        da = torch.zeros_like(a) if (H % 2 or W % 2) else torch.empty_like(a)
        L.check(_k(a).uclstm_maxpool2_bwd(_p(a), _p(dp), None, _p(da), N, H, W, Cp, _stream()), "maxpool2_bwd")
        return da


class MaxPool2Skip(torch.autograd.Function):
    """MaxPool2d(2) of ``a`` plus ``a`` itself as a second output (the skip connection): in the UNet every encoder output is
    used twice (train/unet.py:166-169), so its gradient is a sum of two tensors -- added inside the max-pool backward kernel
    instead of by autograd's separate elementwise add (one read + one write of the full-resolution tensor less)."""

    @staticmethod
    def forward(ctx, a):
        _dev(a, ACT, "activation")
        N, H, W, Cp = a.shape
        p = torch.empty((N, H // 2, W // 2, Cp), dtype=a.dtype, device=a.device)
        L.check(_k(a).uclstm_maxpool2_fwd(_p(a), _p(p), N, H, W, Cp, _stream()), "maxpool2_fwd")
        ctx.save_for_backward(a)
        return p, a.view_as(a)

    @staticmethod
    def backward(ctx, dp, dskip):
        (a,) = ctx.saved_tensors
        N, H, W, Cp = a.shape
        if dp is None:
            return dskip
        dp = dp.contiguous()
        odd = bool(H % 2 or W % 2)
        da = torch.zeros_like(a) if odd else torch.empty_like(a)
        fused = dskip is not None and not odd
        L.check(_k(a).uclstm_maxpool2_bwd(_p(a), _p(dp), _p(dskip.contiguous()) if fused else None, _p(da), N, H, W, Cp, _stream()),
                "maxpool2_bwd")
        if dskip is not None and not fused:
            da = da + dskip
        return da


# ---------------------------------------------------------------------------------------------
# ConvTranspose2d(k=2, s=2)  (train/unet.py:90, :94)
# ---------------------------------------------------------------------------------------------
class ConvT2x2(_GradAwareFunction):
    @staticmethod
    def forward(ctx, x, weight, bias):
        _dev(x, ACT, "activation")
        Ci, Co = weight.shape[0], weight.shape[1]
        Cop = cpad(Co)
        N, h, w, _ = x.shape
        pd = convt_pack_desc(Ci, Co)
        wp = pack_weights(pd, weight, 0, x.dtype)
        bp = pack_bias(pd, bias) if bias is not None else None
        u = torch.empty((N, 2 * h, 2 * w, Cop), dtype=x.dtype, device=x.device)
        segs = [(u, t * Cop, (t + 1) * Cop, 0, 2, t // 2, t % 2) for t in range(4)]
        igemm_store([SrcView(x)], wp, (h, w), N, segs, ktap=1, pad=0, bias=bp)
        ctx.save_for_backward(x, weight, bias)
        ctx.has_bias = bias is not None
        if _will_backward(ctx):
            note_use(weight, bias)
        return u

    @staticmethod
    def backward(ctx, du):
        x, weight, bias = ctx.saved_tensors
        du = du.contiguous()
        Ci, Co = weight.shape[0], weight.shape[1]
        Cop = cpad(Co)
        N, h, w, _ = x.shape
        pd = convt_pack_desc(Ci, Co)
        segs = [(du, t * Cop, (t + 1) * Cop, 0, 2, t // 2, t % 2) for t in range(4)]
        dweight = wgrad_into_param(weight, pd, [x, du], lambda: igemm_wgrad([SrcView(x)], segs, pd.N, pd.Ktot, (h, w), N, ktap=1, pad=0))
        dbias = bias_grad_from_colsum(du, bias, Co) if ctx.has_bias else None
        dx = None
        if ctx.needs_input_grad[0]:
            dd = convt_dgrad_pack_desc(Ci, Co)
            wd = pack_weights(dd, weight, 0, du.dtype)
            dx = torch.empty_like(x)
            igemm_store([SrcView(du)], wd, (h, w), N, [(dx, 0, dd.N, 0, 1, 0, 0)], ktap=2, scale=2, pad=0)
        return dx, dweight, dbias


# ---------------------------------------------------------------------------------------------
# OutConv 1x1  (train/unet.py:101-107)
# ---------------------------------------------------------------------------------------------
class OutConv1x1(_GradAwareFunction):
    """16-bit NHWC in, f32 NCHW out (the model's public output dtype/layout)."""

    @staticmethod
    def forward(ctx, a, weight, bias):
        _dev(a, ACT, "activation")
        N, H, W, Cp = a.shape
        Co, Ci = weight.shape[0], weight.shape[1]
        y = torch.empty((N, Co, H, W), dtype=F32, device=a.device)
        L.check(_k(a).uclstm_outconv_fwd(_p(a), _p(weight), _p(bias), _p(y), N, H * W, Cp, Ci, Co, _stream()), "outconv_fwd")
        ctx.save_for_backward(a, weight, bias)
        ctx.has_bias = bias is not None
        if _will_backward(ctx):
            note_use(weight, bias)
        return y

    @staticmethod
    def backward(ctx, dy):
        a, weight, bias = ctx.saved_tensors
        dy = dy.contiguous().float()
        N, H, W, Cp = a.shape
        Co, Ci = weight.shape[0], weight.shape[1]
        da = torch.empty_like(a) if ctx.needs_input_grad[0] else None
        # the kernel ADDS its block sums into dw / db: with attached f32 gradients it adds straight into them (no zero-fill,
        # no accumulate kernels); the weight is a view [Co, Ci] of the parameter [Co, Ci, 1, 1], so is its gradient
        gw = direct_grad(weight)
        gb = direct_grad(bias) if ctx.has_bias else None
        direct = gw is not None and (gb is not None or not ctx.has_bias) and gw.is_contiguous()
        dw = gw if direct else torch.zeros((Co, Ci), dtype=F32, device=a.device)
        db = gb if (direct and ctx.has_bias) else torch.zeros((Co,), dtype=F32, device=a.device)
        L.check(_k(a).uclstm_outconv_bwd(_p(a), _p(weight), _p(dy), _p(da), _p(dw), _p(db), N, H * W, Cp, Ci, Co, _stream()),
                "outconv_bwd")
        if direct:
            grad_written(weight)
            if ctx.has_bias:
                grad_written(bias)
            return da, None, None
        return da, dw.view_as(weight), (db if ctx.has_bias else None)


# ---------------------------------------------------------------------------------------------
# SpatialAttention  (train/unet.py:113-125)
# ---------------------------------------------------------------------------------------------
class SpatialAttn(_GradAwareFunction):
    """x * sigmoid(conv_kxk([mean_c x, max_c x])) on NHWC 16-bit activations; ``weight`` is the reference's [1,2,k,k] f32."""

    @staticmethod
    def forward(ctx, x, weight, channels):
        _dev(x, ACT, "activation")
        _dev(weight, F32, "attention weight")
        N, H, W, Cp = x.shape
        k = weight.shape[-1]
        dev = x.device
        out = torch.empty_like(x)
        att = torch.empty((N, H, W), dtype=F32, device=dev)
        desc = torch.empty((N, H, W, 2), dtype=F32, device=dev)
        arg = torch.empty((N, H, W), dtype=torch.int32, device=dev)
        L.check(_k(x).uclstm_attention_fwd(_p(x), _p(weight), _p(out), _p(att), _p(desc), _p(arg), N, H, W, Cp, channels, k, _stream()),
                "attention_fwd")
        ctx.save_for_backward(x, weight, att, desc, arg)
        ctx.channels = channels
        if _will_backward(ctx):
            note_use(weight)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, weight, att, desc, arg = ctx.saved_tensors
        dout = dout.contiguous()
        N, H, W, Cp = x.shape
        k = weight.shape[-1]
        dx = torch.empty_like(x)
        scratch = torch.empty((3 * N * H * W + 2,), dtype=F32, device=x.device)
        g = direct_grad(weight)
        dw = g if g is not None else torch.empty_like(weight)
        L.check(_k(x).uclstm_attention_bwd(_p(x), _p(dout), _p(weight), _p(att), _p(desc), _p(arg), _p(dx), _p(dw), int(g is not None),
                                           _p(scratch), N, H, W, Cp, ctx.channels, k, _stream()), "attention_bwd")
        if g is not None:
            grad_written(weight)
            return dx, None, None
        return dx, dw, None


# ---------------------------------------------------------------------------------------------
# ConvLSTM layer over a whole sequence  (train/unet.py:21-36 x T, :55-57)
# ---------------------------------------------------------------------------------------------
class ConvLSTMSeq(torch.autograd.Function):
    """All T steps of one ConvLSTM layer.

    x_all bf16 [T,B,H,W,Cxp]; h0 bf16 [B,H,W,Hdp] / c0 f32 or None (zero state).
    Returns (h_all bf16 [T,B,H,W,Hdp], c_T f32 [B,H,W,Hdp]); h_T is h_all[T-1].
    Forward: one fused kernel per step (gate conv over (x_t, h_{t-1}) + nonlinearities + cell
    update), post-activation gates kept in bf16 for the backward pass when training.
    Backward: per step (reverse) a point-wise kernel + the h-part input-gradient GEMM; the x-part
    input gradient and the weight gradient are batched over all T after the loop.
    """

    @staticmethod
    def forward(ctx, x_all, h0, c0, weight, bias, Hd, Cx, need_grad, out=None):
        ctx.set_materialize_grads(False)
        _dev(x_all, ACT, "x_all")
        adt = x_all.dtype
        K = _k(x_all)
        T, B, H, W, Cxp = x_all.shape
        Hdp = cpad(Hd)
        dev = x_all.device
        ks = weight.shape[-1]
        pd = lstm_pack_desc(Hd, Cx, ks)
        bp = pack_bias(pd, bias) if bias is not None else None
        if out is not None:
            # streaming inference: the single step writes the caller's state buffers (h_out must not be h0: the convolution
            # reads h0's neighbourhoods while h_out is written; c_out may be c0 itself, the cell update is element-wise)
            if need_grad or T != 1 or h0 is None:
                raise L.UclstmError("ConvLSTMSeq: out= is for one-step inference with a carried state")
            for t_, dt_ in ((out[0], adt), (out[1], F32)):
                if not (t_.is_contiguous() and t_.dtype == dt_ and tuple(t_.shape) == (B, H, W, Hdp)):
                    raise L.UclstmError("ConvLSTMSeq: out buffers must be contiguous [B,H,W,Hd_p] (h: activation dtype, c: f32)")
            if out[0].data_ptr() == h0.data_ptr():
                raise L.UclstmError("ConvLSTMSeq: h_out aliases h0")
        h_hist = torch.empty((T + 1, B, H, W, Hdp), dtype=adt, device=dev) if out is None else (None, out[0])
        c_hist = torch.empty((T + 1, B, H, W, Hdp), dtype=F32, device=dev) if out is None else (None, out[1])
        # inference (nothing saved): step 0 reads the caller's state tensors where they lie instead of copies in slot 0
        direct = (not need_grad and h0 is not None and h0.is_contiguous() and h0.dtype == adt and tuple(h0.shape) == (B, H, W, Hdp)
                  and (c0 is None or (c0.is_contiguous() and c0.dtype == F32 and tuple(c0.shape) == (B, H, W, Hdp))))
        if out is not None and not direct:
            raise L.UclstmError("ConvLSTMSeq: out= needs the carried state as contiguous [B,H,W,Hd_p] tensors (h: activation dtype, c: f32)")
        if h0 is None:
            h_hist[0].zero_()
        elif not direct:
            h_hist[0].copy_(h0)
        if c0 is not None and not direct:
            c_hist[0].copy_(c0)
        gates = torch.empty((T, B, H, W, 4, Hdp), dtype=adt, device=dev) if need_grad else None
        pixels = B * H * W
        # Per-step GEMM M = B*H*W.  When its tile grid cannot fill the chip (bottleneck LSTM: M = 512), run the gate
        # convolution as split-K partial tiles in f32 slabs and apply the cell update in a point-wise kernel; otherwise one
        # fused kernel per step (gates never leave registers).
        hoist = HOIST_X and T >= 2
        if hoist:
            # W_x * x_t for ALL timesteps as one GEMM over T*B*H*W pixels (f32 pre-activations in panel-row order); the
            # recurrence then multiplies only by W_h: K and the weight bytes re-read per step halve
            wx = pack_weights(lstm_half_pack_desc(Hd, Cx, "x", ks), weight, 0, adt)
            wp = pack_weights(lstm_half_pack_desc(Hd, Cx, "h", ks), weight, 0, adt)
            N = wp.shape[0]
            pre_x = torch.empty((1, T * pixels, N), dtype=F32, device=dev)
            for i0, i1 in _img_chunks(T * B, 1, max(_bytes_per_img(x_all[0]), H * W * N * 4), "convlstm x half"):
                igemm_atomic([SrcView(x_all.view(T * B, H, W, Cxp)[i0:i1])], wx, (H, W), i1 - i0,
                             pre_x[:, i0 * H * W:i1 * H * W], 1, ktap=ks, pad=ks // 2, slabs=True, kind="igemm_fwd_xhoist")
            pre_x = pre_x.view(T, pixels, N)
        else:
            wp = pack_weights(pd, weight, 0, adt)
        ksplit = split_k_factor(pixels, wp.shape[0], wp.shape[1] // 64)
        # one f32 slab per K range (plain stores; the point-wise kernel adds them): no atomics, nothing to zero
        nsl = ksplit_used(wp.shape[1], ksplit, ks) if ksplit > 1 else 0
        pre = torch.empty((nsl, pixels, wp.shape[0]), dtype=F32, device=dev) if ksplit > 1 else None
        for t in range(T):
            c_prev = (c0 if (direct and t == 0) else c_hist[t]) if (c0 is not None or t > 0) else None
            h_prev = h0 if (direct and t == 0) else h_hist[t]
            g_t = gates[t] if need_grad else None
            px_t = pre_x[t] if hoist else None
            srcs = ([] if hoist else [SrcView(x_all[t])]) + [SrcView(h_prev)]
            if hoist and t == 0 and h0 is None:
                # zero initial state (train/unet.py:23-25): W_h * 0 = 0, the step is the point-wise update of W_x * x_0
                L.check(K.uclstm_lstm_fwd_pointwise(None, 0, 0, 0, _p(px_t), _p(bp), _p(c_prev), _p(c_hist[1]), _p(h_hist[1]),
                                                        _p(g_t), pixels, Hdp, _stream()), "lstm_fwd_pointwise")
            elif ksplit > 1:
                igemm_atomic(srcs, wp, (H, W), B, pre, ksplit, ktap=ks, pad=ks // 2, slabs=True)
                L.check(K.uclstm_lstm_fwd_pointwise(_p(pre), nsl, pre.stride(0), 0, _p(px_t), _p(bp), _p(c_prev), _p(c_hist[t + 1]),
                                                        _p(h_hist[t + 1]), _p(g_t), pixels, Hdp, _stream()), "lstm_fwd_pointwise")
            else:
                igemm_lstm(None if hoist else x_all[t], h_prev, wp, bp, c_prev, c_hist[t + 1], h_hist[t + 1], g_t, ks, pre_add=px_t)
        if need_grad:
            ctx.save_for_backward(x_all, weight, h_hist, c_hist, gates, bias)
            ctx.cfg = (Hd, Cx, c0 is not None, bias is not None, ks)
            note_use(weight, bias)
        if out is not None:
            return out[0].unsqueeze(0), out[1]
        return h_hist[1:], c_hist[T]

    @staticmethod
    def backward(ctx, dh_all, dc_T):
        x_all, weight, h_hist, c_hist, gates, bias = ctx.saved_tensors
        Hd, Cx, has_c0, has_bias, ks = ctx.cfg
        T, B, H, W, Cxp = x_all.shape
        Hdp = cpad(Hd)
        dev = x_all.device
        pixels = B * H * W
        dh_all = None if dh_all is None else dh_all.contiguous()
        adt = x_all.dtype
        K = _k(x_all)
        dgates = torch.empty((T, B, H, W, 4 * Hdp), dtype=adt, device=dev)
        dc = torch.empty((B, H, W, Hdp), dtype=F32, device=dev)
        dc_zero = dc_T is None
        if not dc_zero:
            dc.copy_(dc_T)
        ddh = lstm_dgrad_pack_desc(Hd, Cx, Hd, ks)
        wd_h = pack_weights(ddh, weight, Cx * ks * ks, adt)
        dh_rec = None
        # recurrent gradient dh_{t-1} = W_h^T (*) dgates_t: M = B*H*W pixels, N = Hd, K = 9*4*Hd.  Small M -> split-K with
        # f32 atomics into dh (read back as f32 by the next step's point-wise kernel); else a plain bf16 store.
        ksplit = split_k_factor(pixels, ddh.N, ddh.Ktot // 64)
        # split-K: one f32 slab per K range, plain stores; the point-wise kernel of the next (earlier) timestep adds them.
        # Two buffers alternate so that step t's GEMM never writes what step t+1's point-wise kernel still reads.
        nsl = ksplit_used(ddh.Ktot, ksplit, ks) if ksplit > 1 else 1
        if ksplit > 1:
            buf = [torch.empty((nsl, B, H, W, Hdp), dtype=F32, device=dev) for _ in range(2)]
        else:
            buf = [torch.empty((B, H, W, Hdp), dtype=adt, device=dev) for _ in range(2)]
        need_h0 = ctx.needs_input_grad[1]
        for t in range(T - 1, -1, -1):
            c_prev = c_hist[t] if (has_c0 or t > 0) else None
            L.check(K.uclstm_lstm_bwd_pointwise(_p(gates[t]), _p(c_prev), _p(c_hist[t + 1]),
                                                    _p(dh_all[t]) if dh_all is not None else None, _p(dh_rec),
                                                    1 if ksplit > 1 else 0, nsl, pixels * Hdp, _p(dc),
                                                    int(dc_zero), _p(dgates[t]), pixels, Hdp, _stream()), "lstm_bwd_pointwise")
            dc_zero = False
            if t > 0 or need_h0:
                dh_rec = buf[t & 1]
                if ksplit > 1:
                    igemm_atomic([SrcView(dgates[t])], wd_h, (H, W), B, dh_rec.view(nsl, pixels, Hdp), ksplit, ktap=ks, pad=ks // 2,
                                 slabs=True)
                else:
                    igemm_store([SrcView(dgates[t])], wd_h, (H, W), B, [(dh_rec, 0, ddh.N, 0, 1, 0, 0)], ktap=ks, pad=ks // 2)
        dg_flat = dgates.view(T * B, H, W, 4 * Hdp)
        x_flat = x_all.reshape(T * B, H, W, Cxp)
        hprev_flat = h_hist[:T].reshape(T * B, H, W, Hdp)
        ud = lstm_wgrad_unpack_desc(Hd, Cx, ks)
        dweight = wgrad_into_param(weight, ud, [x_all, h_hist, dgates],
                                   lambda: igemm_wgrad([SrcView(x_flat), SrcView(hprev_flat)], [(dg_flat, 0, 4 * Hdp, 0, 1, 0, 0)],
                                                       ud.N, ud.Ktot, (H, W), T * B, ktap=ks, pad=ks // 2))
        dbias = None
        if has_bias:
            if Hdp == Hd:           # dgates columns are (gate, hidden channel) = the bias order of train/unet.py:29
                dbias = bias_grad_from_colsum(dg_flat, bias, 4 * Hd)
            else:
                dbias = colsum(dg_flat).view(4, Hdp)[:, :Hd].reshape(4 * Hd).contiguous()
        dx_all = None
        if ctx.needs_input_grad[0]:
            ddx = lstm_dgrad_pack_desc(Hd, Cx, Cx, ks)
            wd_x = pack_weights(ddx, weight, 0, adt)
            dx_all = torch.empty_like(x_all)
            igemm_store([SrcView(dg_flat)], wd_x, (H, W), T * B, [(dx_all.view(T * B, H, W, Cxp), 0, ddx.N, 0, 1, 0, 0)], ktap=ks,
                        pad=ks // 2)
        dh0 = (dh_rec if dh_rec.dtype == adt else dh_rec.sum(dim=0).to(adt)) if need_h0 else None
        dc0 = dc if (has_c0 and ctx.needs_input_grad[2]) else None
        return dx_all, dh0, dc0, dweight, dbias, None, None, None


# ---------------------------------------------------------------------------------------------
# Several independent ConvLSTMs advancing in lockstep (train/unet.py:185-191 runs temporal, lstm_skip3, lstm_skip2 one after
# the other; they do not depend on each other)
# ---------------------------------------------------------------------------------------------
GROUP_LSTM = os.environ.get("UCLSTM_GROUP_LSTM", "1") != "0"
FUSE_POOL = os.environ.get("UCLSTM_FUSE_POOL", "1") != "0"          # MaxPool2d(2) fused into the BatchNorm stage that feeds it (training)
FUSE_HEAD = os.environ.get("UCLSTM_FUSE_HEAD", "1") != "0"          # 1x1 output convolution fused into the last BatchNorm stage (training)
_GROUP_PLANS: dict = {}
_CUS = 256
_BLOCK_OVERHEAD_STEPS = 12.0          # prologue + epilogue of a patch-shape block in units of one K-step (~10 us / 0.87 us)


def _lpt_makespan(blocks: List[Tuple[float, int]], cus: int = _CUS) -> float:
    """Makespan of (duration, count) block classes handed out longest first to ``cus`` one-block-at-a-time workers."""
    import heapq
    free = [0.0] * cus
    for dur, cnt in sorted(blocks, reverse=True):
        for _ in range(cnt):
            t = heapq.heappop(free)
            heapq.heappush(free, t + dur)
    return max(free)


def plan_group_ksplit(members: Sequence[Tuple[int, int]]) -> Tuple[int, ...]:
    """K ranges per member of a group launch.  ``members``: (tiles, chunks) = 128 x 256 output tiles and (source, 64-channel)
    K chunks of each member's GEMM; a block of member i runs ceil(chunks_i / ksplit_i) chunks of 9 K-steps plus a fixed
    prologue / epilogue.  Exhaustive search over ksplit <= 16 per member for the smallest longest-block-first makespan on the
    256 CUs (every extra K range also costs an f32 slab round trip, priced at a quarter of a K-step per slab and tile row)."""
    key = tuple(members)
    hit = _GROUP_PLANS.get(key)
    if hit is not None:
        return hit
    import itertools
    options = []
    for tiles, chunks in members:
        ks = sorted({k for k in range(1, min(chunks, 16) + 1) if (chunks + k - 1) // k != (chunks + k - 2) // max(k - 1, 1) or k == 1})
        options.append(ks)
    best, best_cost = None, None
    for combo in itertools.product(*options):
        blocks, slab_cost = [], 0.0
        for (tiles, chunks), k in zip(members, combo):
            cper = (chunks + k - 1) // k
            used = (chunks + cper - 1) // cper
            full, rem = divmod(chunks, cper)
            blocks.append((9.0 * cper + _BLOCK_OVERHEAD_STEPS, tiles * full))
            if rem:
                blocks.append((9.0 * rem + _BLOCK_OVERHEAD_STEPS, tiles))
            if used > 1:
                slab_cost += 0.25 * used * tiles / _CUS * 8
        cost = _lpt_makespan(blocks) + slab_cost
        if best_cost is None or cost < best_cost - 1e-9:
            best, best_cost = combo, cost
    _GROUP_PLANS[key] = best
    return best


class _LstmMember:
    """Per-member state of ConvLSTMGroup (what ConvLSTMSeq keeps in locals)."""
    pass


def convlstm_group_forward(members, need_grad: bool):
    """All T forward steps of n independent single-layer ConvLSTMs with ONE group launch per timestep (uclstm_igemm_fwd_group)
    instead of n launches.  ``members``: (x_all [T,B,H,W,Cxp], h0, c0, weight, bias, Hd, Cx) each; returns per member
    (h_hist [T+1,...], c_hist [T+1,...], gates or None) -- exactly what ConvLSTMSeq.forward keeps -- so that every member gets
    its OWN autograd node (ConvLSTMSeqPre) and its backward pass runs when ITS gradient arrives, as with separate sequences.
    (One node for the whole group was measured: backward 21.3 -> 21.8 ms, because the three weight-gradient GEMMs then reach
    the side stream together at the end instead of each overlapping the next LSTM's backward recurrence.)
    Same arithmetic as ConvLSTMSeq: a member's step is the fused cell kernel or split-K slabs + the point-wise kernel; only the
    K-range counts are planned for the group (plan_group_ksplit), i.e. results agree to f32 summation order of the K ranges."""
    ms = []
    for x_all, h0, c0, weight, bias, Hd, Cx in members:
        m = _LstmMember()
        m.x_all, m.h0, m.c0, m.weight, m.bias, m.Hd, m.Cx = x_all, h0, c0, weight, bias, Hd, Cx
        _dev(m.x_all, ACT, "x_all")
        m.T, m.B, m.H, m.W, m.Cxp = m.x_all.shape
        m.Hdp = cpad(Hd)
        m.pixels = m.B * m.H * m.W
        m.pd = lstm_pack_desc(Hd, Cx, 3)
        m.bp = pack_bias(m.pd, m.bias) if m.bias is not None else None
        m.wp = pack_weights(m.pd, m.weight, 0, m.x_all.dtype)
        ms.append(m)
    T, adt, dev = ms[0].T, ms[0].x_all.dtype, ms[0].x_all.device
    K = L.kernels(adt)
    plan = plan_group_ksplit([(((m.wp.shape[0] + 127) // 128) * (m.pixels // 256), m.wp.shape[1] // (64 * 9)) for m in ms])
    for m, ks in zip(ms, plan):
        m.h_hist = torch.empty((T + 1, m.B, m.H, m.W, m.Hdp), dtype=adt, device=dev)
        m.c_hist = torch.empty((T + 1, m.B, m.H, m.W, m.Hdp), dtype=F32, device=dev)
        if m.h0 is None:
            m.h_hist[0].zero_()
        else:
            m.h_hist[0].copy_(m.h0)
        if m.c0 is not None:
            m.c_hist[0].copy_(m.c0)
        m.gates = torch.empty((T, m.B, m.H, m.W, 4, m.Hdp), dtype=adt, device=dev) if need_grad else None
        m.ksplit = ks
        m.nsl = ksplit_used(m.wp.shape[1], ks, 3) if ks > 1 else 0
        m.pre = torch.empty((m.nsl, m.pixels, m.wp.shape[0]), dtype=F32, device=dev) if ks > 1 else None
    for t in range(T):
        items, pws = [], []
        for m in ms:
            c_prev = m.c_hist[t] if (m.c0 is not None or t > 0) else None
            g_t = m.gates[t] if need_grad else None
            if m.ksplit > 1:
                items.append(_atomic_desc([SrcView(m.x_all[t]), SrcView(m.h_hist[t])], m.wp, (m.H, m.W), m.B, m.pre, m.ksplit, ktap=3, pad=1,
                                          slabs=True))
                a = L.LstmFwdPwArgs()
                a.pre, a.nslab, a.slab, a.clear, a.pre_add = m.pre.data_ptr(), m.nsl, m.pre.stride(0), 0, None
                a.bias = None if m.bp is None else m.bp.data_ptr()
                a.c_prev = None if c_prev is None else c_prev.data_ptr()
                a.c_out, a.h_out = m.c_hist[t + 1].data_ptr(), m.h_hist[t + 1].data_ptr()
                a.gates_out = None if g_t is None else g_t.data_ptr()
                a.pixels, a.Hd_p = m.pixels, m.Hdp
                pws.append(a)
            else:
                items.append(_lstm_desc(m.x_all[t], m.h_hist[t], m.wp, m.bp, c_prev, m.c_hist[t + 1], m.h_hist[t + 1], g_t, 3))
        igemm_group(items, K)
        if pws:
            arr = (L.LstmFwdPwArgs * len(pws))(*pws)
            L.check(K.uclstm_lstm_fwd_pointwise_group(arr, len(pws), _stream()), "lstm_fwd_pointwise_group")
    return [(m.h_hist, m.c_hist, m.gates) for m in ms]


def convlstm_group_step(members) -> bool:
    """ONE inference step of n independent single-layer ConvLSTMs with carried state as one group launch (streaming rollout,
    BASELINE configs[4]).  ``members``: (x_t [B,H,W,Cxp], h_prev, c_prev, h_out, c_out, weight, bias, Hd, Cx); the new state is
    written into h_out / c_out (h_out must not alias h_prev; c_out may be c_prev).  Returns False -- nothing launched -- when the
    members cannot form a group (the caller then steps them one by one)."""
    if not GROUP_LSTM or not 2 <= len(members) <= 4:
        return False
    ms = []
    for x_t, h_prev, c_prev, h_out, c_out, weight, bias, Hd, Cx in members:
        if (weight.shape[-1] != 3 or cpad(Hd) != Hd or Hd % 64 or x_t.shape[3] % 64 or (x_t.shape[0] * x_t.shape[1] * x_t.shape[2]) % 256
                or h_prev is None or c_prev is None or h_out.data_ptr() == h_prev.data_ptr() or x_t.dtype != members[0][0].dtype):
            return False
        m = _LstmMember()
        m.x_t, m.h_prev, m.c_prev, m.h_out, m.c_out, m.Hd = x_t, h_prev, c_prev, h_out, c_out, Hd
        m.B, m.H, m.W, _ = x_t.shape
        m.pixels = m.B * m.H * m.W
        m.pd = lstm_pack_desc(Hd, Cx, 3)
        m.bp = pack_bias(m.pd, bias) if bias is not None else None
        m.wp = pack_weights(m.pd, weight, 0, x_t.dtype)
        ms.append(m)
    adt, dev = ms[0].x_t.dtype, ms[0].x_t.device
    K = L.kernels(adt)
    plan = plan_group_ksplit([(((m.wp.shape[0] + 127) // 128) * (m.pixels // 256), m.wp.shape[1] // (64 * 9)) for m in ms])
    items, pws, keep = [], [], []
    for m, ks in zip(ms, plan):
        if ks > 1:
            nsl = ksplit_used(m.wp.shape[1], ks, 3)
            pre = torch.empty((nsl, m.pixels, m.wp.shape[0]), dtype=F32, device=dev)
            keep.append(pre)
            items.append(_atomic_desc([SrcView(m.x_t), SrcView(m.h_prev)], m.wp, (m.H, m.W), m.B, pre, ks, ktap=3, pad=1, slabs=True))
            a = L.LstmFwdPwArgs()
            a.pre, a.nslab, a.slab, a.clear, a.pre_add = pre.data_ptr(), nsl, pre.stride(0), 0, None
            a.bias = None if m.bp is None else m.bp.data_ptr()
            a.c_prev, a.c_out, a.h_out, a.gates_out = m.c_prev.data_ptr(), m.c_out.data_ptr(), m.h_out.data_ptr(), None
            a.pixels, a.Hd_p = m.pixels, m.Hd
            pws.append(a)
        else:
            items.append(_lstm_desc(m.x_t, m.h_prev, m.wp, m.bp, m.c_prev, m.c_out, m.h_out, None, 3))
    if not group_launchable([it[0] for it in items]):
        return False
    igemm_group(items, K)
    if pws:
        arr = (L.LstmFwdPwArgs * len(pws))(*pws)
        L.check(K.uclstm_lstm_fwd_pointwise_group(arr, len(pws), _stream()), "lstm_fwd_pointwise_group")
    return True


class ConvLSTMSeqPre(torch.autograd.Function):
    """ConvLSTMSeq whose forward pass has already been computed (convlstm_group_forward): the node only records what
    ConvLSTMSeq.forward would have saved; its backward IS ConvLSTMSeq.backward."""

    @staticmethod
    def forward(ctx, x_all, h0, c0, weight, bias, Hd, Cx, need_grad, h_hist, c_hist, gates):
        ctx.set_materialize_grads(False)
        T = x_all.shape[0]
        if need_grad:
            ctx.save_for_backward(x_all, weight, h_hist, c_hist, gates, bias)
            ctx.cfg = (Hd, Cx, c0 is not None, bias is not None, weight.shape[-1])
            note_use(weight, bias)
        return h_hist[1:], c_hist[T]

    @staticmethod
    def backward(ctx, dh_all, dc_T):
        return ConvLSTMSeq.backward(ctx, dh_all, dc_T) + (None, None, None)


def convlstm_group_ok(members) -> bool:
    """Can these (x_all, h0, c0, weight, bias, Hd, Cx) single-layer ConvLSTMs run as ConvLSTMGroup?  Every per-step GEMM must
    take the patch shape (3x3 gate convolution, channel counts multiples of 64, B*h*w a multiple of 256, ...) -- checked with the
    library's own planner on the step-0 descriptors -- and all members share T, B and the activation dtype."""
    if not GROUP_LSTM or len(members) < 2 or len(members) > 4:
        return False
    x0 = members[0][0]
    try:
        descs = []
        for x_all, h0, c0, weight, bias, Hd, Cx in members:
            if x_all.dim() != 5 or x_all.shape[0] != x0.shape[0] or x_all.shape[1] != x0.shape[1] or x_all.dtype != x0.dtype:
                return False
            if weight.shape[-1] != 3 or cpad(Hd) != Hd or Hd % 64 or x_all.shape[4] % 64 or (x_all.shape[1] * x_all.shape[2] * x_all.shape[3]) % 256:
                return False
            T, B, H, W, Cxp = x_all.shape
            pd = lstm_pack_desc(Hd, Cx, 3)
            # a descriptor with the right shapes (the planner does not dereference the pointers): x_t, h, panel geometry
            d = L.IgemmDesc()
            d.n_img, d.H, d.W, d.groups = B, H, W, 1
            d.ktap, d.scale, d.pad, d.nsrc = 3, 1, 1, 2
            for i, cch in enumerate((Cxp, Hd)):
                d.src[i].ptr, d.src[i].C, d.src[i].Hs, d.src[i].Ws = x_all.data_ptr(), cch, H, W
            d.wp, d.N, d.Ktot = x_all.data_ptr(), pd.N, pd.Ktot
            d.epi, d.Hd_p = L.EPI_LSTM, Hd
            d.c_out = d.h_out = x_all.data_ptr()
            descs.append(d)
        return group_launchable(descs)
    except Exception:
        return False


# ---------------------------------------------------------------------------------------------
# Loss (main.py:28-72)
# ---------------------------------------------------------------------------------------------
class LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, y_pred, y, mask, use_mask):
        _dev(y_pred, F32, "y_pred")
        y = _dev(y.contiguous(), F32, "y")
        H, W = y_pred.shape[-2], y_pred.shape[-1]
        planes = y_pred.numel() // (H * W)
        m = _dev(mask.contiguous().float(), F32, "mask") if (use_mask and mask is not None) else None
        sums = torch.zeros((4,), dtype=torch.float64, device=y_pred.device)
        L.check(L.lib.uclstm_loss_fwd(_p(y_pred), _p(y), _p(m), _p(sums), planes, H, W, _stream()), "loss_fwd")
        n1 = float(y_pred.numel())
        n2 = float(planes * (H - 1) * (W - 1))
        if m is not None:
            d1 = sums[1] + 1e-8
            d2 = sums[3] + 1e-8
        else:
            d1 = torch.full((), n1, dtype=torch.float64, device=y_pred.device)
            d2 = torch.full((), n2, dtype=torch.float64, device=y_pred.device)
        loss = (sums[0] / d1 + 0.005 * sums[2] / d2).float()
        ctx.save_for_backward(y_pred, y, m, (1.0 / d1).float(), (0.005 / d2).float())
        return loss

    @staticmethod
    def backward(ctx, g):
        y_pred, y, m, c1, c2 = ctx.saved_tensors
        H, W = y_pred.shape[-2], y_pred.shape[-1]
        planes = y_pred.numel() // (H * W)
        grad = torch.empty_like(y_pred)
        coefs = torch.stack((c1 * g, c2 * g)).float().contiguous()      # stays on device: no host sync in the step
        L.check(L.lib.uclstm_loss_bwd(_p(y_pred), _p(y), _p(m), _p(coefs), _p(grad), planes, H, W, _stream()), "loss_bwd")
        return grad, None, None, None
