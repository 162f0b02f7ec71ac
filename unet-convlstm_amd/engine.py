"""Training / evaluation step loop of the reference's main.py on the HIP path, plus data sources.

``train_one_epoch`` / ``evaluate`` keep the reference signatures and return values
(main.py:77-144, :150-205: ``(avg_loss, mae, rmse, mean_err)`` in de-normalised units) but keep
the metric block on the device as running sums instead of per-pixel Python lists
(main.py:114-133), and do not synchronise with the host inside the step.
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch

from . import ops
from .loss import compute_loss
from .optim import FusedAdamW


# ---------------------------------------------------------------------------------------------
# data
# ---------------------------------------------------------------------------------------------
class SyntheticSequences:
    """Seeded synthetic batches in the post-normalisation ranges of the reference dataset
    (X ~ U[0,1) as after ``x / norm_const``, train/unet.py:283; Y ~ U(-1,1) as
    after the [-1,1] mapping, ``:299``; mask = cloud pixels).  ``kind='blobs'`` draws moving blobs with
    integer velocities in [-5,5] and the bounce rule of digits/build_moving_mnist.py:38-47; X is the frame
    duplicated into both "satellite" channels and Y the per-pixel vx map."""

    def __init__(self, B: int, T: int, H: int, W: int, seed: int = 1, kind: str = "uniform", device="cuda", channels: int = 2):
        g = torch.Generator(device="cpu").manual_seed(seed)
        if kind == "uniform":
            x = torch.rand((B, T, channels, H, W), generator=g)
            y = torch.rand((B, T, 1, H, W), generator=g) * 2 - 1
            mask = (torch.rand((B, T, 1, H, W), generator=g) > 0.3).float()
        elif kind == "blobs":
            x = torch.zeros((B, T, channels, H, W))
            y = torch.zeros((B, T, 1, H, W))
            yy, xx = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
            for b in range(B):
                for _ in range(2):
                    r = 4 + int(torch.randint(0, 4, (1,), generator=g))
                    px = float(torch.randint(r, W - r, (1,), generator=g))
                    py = float(torch.randint(r, H - r, (1,), generator=g))
                    vx = int(torch.randint(-5, 6, (1,), generator=g))
                    vy = int(torch.randint(-5, 6, (1,), generator=g))
                    for t in range(T):
                        # bump 1/(1+q)^2 in f64 from IEEE basic operations only (+, *, /): bit-identical on every host CPU
                        # (exp() differs by an ulp between vector ISAs, which breaks seed-regenerated fixtures)
                        q = ((xx - px).double() ** 2 + (yy - py).double() ** 2) / (2.0 * (r / 2.0) ** 2)
                        blob = (1.0 / ((1.0 + q) * (1.0 + q))).float()
                        on = blob > 0.1
                        x[b, t, :, on] = torch.maximum(x[b, t, :, on], blob[on])
                        y[b, t, 0][on] = vx / 5.0
                        px, py = px + vx, py + vy
                        if px < r or px > W - 1 - r:
                            vx = -vx
                            px = min(max(px, r), W - 1 - r)
                        if py < r or py > H - 1 - r:
                            vy = -vy
                            py = min(max(py, r), H - 1 - r)
            mask = (x[:, :, 0:1] > 0.1).float()
        else:
            raise ValueError(kind)
        self.x, self.y, self.mask = x.to(device), y.to(device), mask.to(device)

    def __iter__(self):
        yield self.x, self.y, self.mask

    def __len__(self):
        return 1


class NPZSequenceDataset(torch.utils.data.Dataset):
    """Host-side mirror of the reference dataset (train/unet.py:210-327): ``.npz`` with ``X [N,T,2,H,W]``,
    ``Y [N,T,1,H,W]``; mask from RAW x > 1.1 (``:279``), x / max(x_max, 1) (``:220,:283``), Y clipped, asinh(y/scale)
    and mapped to [-1,1] (``:287-299``).  ``denormalize`` accepts numpy or torch (any device) and stays on the
    input's device for tensors."""

    def __init__(self, npz_path, lower_percentile=0.00001, upper_percentile=99.99999, clip_outliers=True,
                 min_y=-7.5987958908081055, max_y=8.784920692443848, y_transform="asinh", y_transform_scale=None,
                 y_transform_percentile=99):
        with np.load(npz_path, allow_pickle=False) as data:
            self.X = data["X"].astype(np.float32)
            self.Y = data["Y"].astype(np.float32)
        self.N, self.T, _, self.H, self.W = self.X.shape
        self.x_max = float(np.max(self.X))
        self.norm_const = max(self.x_max, 1.0)
        if y_transform not in ("asinh", "signed_log", None, "none"):
            raise ValueError(y_transform)
        self.y_transform = y_transform
        if y_transform_scale is None:
            self.y_scale = float(np.percentile(np.abs(self.Y), y_transform_percentile)) if y_transform_percentile is not None else 1.0
        else:
            self.y_scale = float(y_transform_scale)
        explicit = (min_y is not None) and (max_y is not None)
        if explicit:
            self.min_vel, self.max_vel = float(min_y), float(max_y)
            self.trans_min = float(self._fwd(np.float64(self.min_vel)))
            self.trans_max = float(self._fwd(np.float64(self.max_vel)))
        else:
            self.min_vel = float(np.percentile(self.Y, lower_percentile))
            self.max_vel = float(np.percentile(self.Y, upper_percentile))
            yt = self._fwd(self.Y)
            self.trans_min = float(np.percentile(yt, lower_percentile))
            self.trans_max = float(np.percentile(yt, upper_percentile))
        if self.trans_max == self.trans_min:
            self.trans_max = self.trans_min + 1.0
        self.clip_outliers = clip_outliers

    def _fwd(self, arr):
        if self.y_transform == "asinh":
            return np.arcsinh(arr / self.y_scale)
        if self.y_transform == "signed_log":
            return np.sign(arr) * np.log1p(np.abs(arr) / self.y_scale)
        return arr

    def __len__(self):
        return self.N

    def __getitem__(self, idx):
        x = torch.from_numpy(self.X[idx])
        mask = (x[:, 0:1] > 1.1).float()
        x = x / self.norm_const
        y_raw = self.Y[idx]
        if self.clip_outliers:
            y_raw = np.clip(y_raw, self.min_vel, self.max_vel)
        y_scaled = (2 * (self._fwd(y_raw) - self.trans_min) / (self.trans_max - self.trans_min) - 1.0).astype(np.float32)
        return x, torch.from_numpy(y_scaled), mask

    def denormalize(self, y_norm):
        if isinstance(y_norm, torch.Tensor):
            yt = (y_norm + 1.0) / 2.0 * (self.trans_max - self.trans_min) + self.trans_min
            if self.y_transform == "asinh":
                return torch.sinh(yt) * self.y_scale
            if self.y_transform == "signed_log":
                return torch.sign(yt) * torch.expm1(yt.abs()) * self.y_scale
            return yt
        yt = (y_norm + 1.0) / 2.0 * (self.trans_max - self.trans_min) + self.trans_min
        if self.y_transform == "asinh":
            return np.sinh(yt) * self.y_scale
        if self.y_transform == "signed_log":
            return np.sign(yt) * (np.expm1(np.abs(yt)) * self.y_scale)
        return yt


def device_transform(ds, x_raw: torch.Tensor, y_raw: torch.Tensor):
    """``NPZSequenceDataset.__getitem__`` (train/unet.py:273-304) for a whole RAW batch already on the device:
    ``x_raw [B,T,C,H,W]``, ``y_raw [B,T,1,H,W]`` f32 -> ``(x, y, mask)`` exactly as the host dataset would yield them
    (asinh transform only).  One kernel; lets the loader ship raw ``.npz`` slabs and skip the per-item numpy work."""
    from . import _lib as L
    if ds.y_transform != "asinh":
        raise ValueError("device_transform implements the 'asinh' target transform (the reference default)")
    x_raw = ops._dev(x_raw.contiguous(), torch.float32, "x_raw")
    y_raw = ops._dev(y_raw.contiguous(), torch.float32, "y_raw")
    B, T, Cc, H, W = x_raw.shape
    x, y = torch.empty_like(x_raw), torch.empty_like(y_raw)
    mask = torch.empty_like(y_raw)
    L.check(L.lib.uclstm_dataset_transform(ops._p(x_raw), ops._p(y_raw), ops._p(x), ops._p(y), ops._p(mask), B * T, Cc, H * W,
                                           float(ds.norm_const), float(ds.min_vel), float(ds.max_vel), int(bool(ds.clip_outliers)),
                                           float(ds.y_scale), float(ds.trans_min), float(ds.trans_max), ops._stream()),
            "dataset_transform")
    return x, y, mask


# ---------------------------------------------------------------------------------------------
# step / epoch loops
# ---------------------------------------------------------------------------------------------
def _stack(output):
    pre = getattr(output, "stacked", None)          # modules.SeqList: the frames as one [B,T,...] view of the kernel's output
    if pre is not None:
        return pre
    return torch.stack(output, dim=1) if isinstance(output, (list, tuple)) else output      # main.py:97-100


def train_step(model, optimizer, x, y, mask=None, use_mask=True, ddp=None, clip_norm: Optional[float] = 1.0):
    """zero_grad -> forward -> stack -> loss -> backward -> [gradient all-reduce] -> clip(1.0) -> optimiser step
    (main.py:91-108).  Returns ``(loss, y_pred)`` as device tensors; nothing here waits for the GPU."""
    optimizer.zero_grad(set_to_none=True)
    if ddp is not None:
        ddp.reset()
    on_gpu = x.is_cuda
    if on_gpu:
        from . import ops
        ops.prepack_begin()          # weights are fixed until optimizer.step(): pack this step's panels ahead, off the main stream
    try:
        output, _ = model(x)
        y_pred = _stack(output)
        loss = compute_loss(y_pred, y, mask, use_mask)
        # fp16 compute: backward runs on loss * scale (FusedAdamW(loss_scale=...) owns the device-side dynamic scale)
        (optimizer.scale_loss(loss) if hasattr(optimizer, "scale_loss") else loss).backward()
    finally:
        if on_gpu:
            ops.prepack_end()
    if ddp is not None:
        ddp.finalize()
    if isinstance(optimizer, FusedAdamW):
        optimizer.max_grad_norm = clip_norm          # clip fused into the optimiser kernels (device-side coefficient)
    elif clip_norm is not None:
        torch.nn.utils.clip_grad_norm_(model.parameters(), clip_norm)                         # main.py:106
    optimizer.step()
    return loss.detach(), y_pred.detach()


class GraphedTrainStep:
    """``train_step`` captured ONCE as a HIP graph and replayed: a training step is ~400 kernel launches issued from Python
    through ctypes (12-19 ms of host time); at the per-GPU batch the benchmark uses the device needs longer than that, but at
    small batches (strong scaling: a global batch of 32 over 8 GPUs is 4 sequences each) the step is bound by the host.  A
    replay costs the host one call.

    What makes the step capturable: no host synchronisation anywhere in it; the optimiser reads its hyper-parameters and step
    count from device memory (``FusedAdamW(capturable=True)``); the weight-gradient / BatchNorm side stream forks from and joins
    the capturing stream through events; the look-ahead panel packing uses a persistent job table.  Inputs are copied into
    static buffers; ``loss`` / ``y_pred`` are static outputs (valid until the next call).  Shapes, ``use_mask`` and the model's
    mode are fixed at capture; ``clip_norm`` / lr changes reach the device through ``optimizer.sync_hyper()`` before a replay.
    Data-parallel training keeps the eager step (its collectives are launched from backward hooks)."""

    def __init__(self, model, optimizer, x, y, mask=None, use_mask: bool = True, clip_norm: Optional[float] = 1.0, warmup: int = 3):
        if not (isinstance(optimizer, FusedAdamW) and optimizer.capturable):
            raise ValueError("GraphedTrainStep needs FusedAdamW(..., capturable=True)")
        if not x.is_cuda:
            raise ops.L.UclstmError("GraphedTrainStep: HIP device tensors required")
        self.model, self.optimizer, self.use_mask, self.clip_norm = model, optimizer, use_mask, clip_norm
        self.x, self.y = x.clone(), y.clone()
        self.mask = None if mask is None else mask.clone()
        optimizer.max_grad_norm = clip_norm
        optimizer.sync_hyper()
        # eager warm-up steps on a side stream (the capture runs on one too): the look-ahead packing plan, the stream pair,
        # the caching allocator's pools and every first-launch attribute call exist before the capture
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(max(warmup, 2)):
                train_step(model, optimizer, self.x, self.y, self.mask, use_mask, None, clip_norm)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.loss, self.y_pred = train_step(model, optimizer, self.x, self.y, self.mask, use_mask, None, clip_norm)
        self.replays = 0

    def __call__(self, x, y, mask=None):
        if x is not self.x:
            self.x.copy_(x, non_blocking=True)
        if y is not self.y:
            self.y.copy_(y, non_blocking=True)
        if self.mask is not None and mask is not None and mask is not self.mask:
            self.mask.copy_(mask, non_blocking=True)
        self.optimizer.max_grad_norm = self.clip_norm
        self.optimizer.sync_hyper()
        self.graph.replay()
        self.replays += 1
        ops.weights_changed()          # what optimizer.step() tells the panel caches in an eager step
        return self.loss, self.y_pred


class _Metrics:
    """Running sums of |d|, d^2, d and the count on the device (replaces main.py:114-142)."""

    def __init__(self, device):
        self.s = torch.zeros(4, dtype=torch.float64, device=device)

    @torch.no_grad()
    def add(self, dataset_obj, y, y_pred, mask, use_mask):
        if y_pred.is_cuda and getattr(dataset_obj, "y_transform", None) == "asinh" and hasattr(dataset_obj, "trans_min"):
            # one fused kernel: de-normalise both, difference, masked running sums (no intermediate tensors)
            from . import _lib as L
            yp = y_pred.contiguous().float()
            yt = y.contiguous().float()
            m = mask.contiguous().float() if (use_mask and mask is not None) else None
            L.check(L.lib.uclstm_metric_sums(ops._p(yp), ops._p(yt), ops._p(m), ops._p(self.s), yp.numel(),
                                             float(dataset_obj.y_scale), float(dataset_obj.trans_min), float(dataset_obj.trans_max),
                                             ops._stream()), "metric_sums")
            return
        d = (dataset_obj.denormalize(y_pred) - dataset_obj.denormalize(y)).double()
        if use_mask:
            m = (mask != 0).double()
            self.s += torch.stack(((d.abs() * m).sum(), (d * d * m).sum(), (d * m).sum(), m.sum()))
        else:
            self.s += torch.stack((d.abs().sum(), (d * d).sum(), d.sum(),
                                   torch.tensor(float(d.numel()), dtype=torch.float64, device=d.device)))

    def result(self):
        a, q, e, n = (float(v) for v in self.s.cpu())
        if n <= 0:
            return 0.0, 0.0, 0.0
        return a / n, math.sqrt(q / n), e / n


def quiesce_host_gc() -> None:
    """Collect once and move every object alive now into the permanent generation (``gc.freeze``).

    A training step is ~500 kernel launches from Python; the host runs 1-3 steps ahead of the device.  A full (generation 2)
    garbage collection of a process that has imported torch walks ~10^6 objects and stops the host for ~75 ms (measured,
    ``tools/spike_hunt.py``): whenever the host is less than that ahead -- after any synchronisation: the start of an epoch,
    a ``loss.item()``, a benchmark's timed region -- the device runs dry and ONE step takes 60-130 ms instead of 33.  After
    ``gc.freeze()`` later collections only walk objects created since, which takes microseconds.  Call it once the model,
    the optimiser and the first steps' caches exist; ``train_one_epoch`` does after its second step.
    """
    import gc
    gc.collect()
    gc.freeze()


def train_one_epoch(model, loader, optimizer, device, dataset_obj, use_mask=True, ddp=None):
    """Reference main.py:77-144; returns ``(avg_loss, avg_mae, avg_rmse, avg_me)``."""
    model.train()
    total = torch.zeros((), dtype=torch.float64, device=device)
    n = 0
    met = _Metrics(device)
    for it, (x, y, mask) in enumerate(loader):
        if it == 2:
            quiesce_host_gc()
        x, y, mask = x.to(device, non_blocking=True), y.to(device, non_blocking=True), mask.to(device, non_blocking=True)
        loss, y_pred = train_step(model, optimizer, x, y, mask, use_mask, ddp)
        total += loss.double() * x.size(0)
        n += x.size(0)
        met.add(dataset_obj, y, y_pred, mask, use_mask)
    mae, rmse, me = met.result()
    return float(total) / max(n, 1), mae, rmse, me


@torch.no_grad()
def evaluate(model, loader, device, dataset_obj, use_mask=True):
    """Reference main.py:150-205."""
    model.eval()
    total = torch.zeros((), dtype=torch.float64, device=device)
    n = 0
    met = _Metrics(device)
    for x, y, mask in loader:
        x, y, mask = x.to(device, non_blocking=True), y.to(device, non_blocking=True), mask.to(device, non_blocking=True)
        output, _ = model(x)
        y_pred = _stack(output)
        loss = compute_loss(y_pred, y, mask, use_mask)
        total += loss.double() * x.size(0)
        n += x.size(0)
        met.add(dataset_obj, y, y_pred, mask, use_mask)
    mae, rmse, me = met.result()
    return float(total) / max(n, 1), mae, rmse, me
