"""Flat-buffer AdamW with fused global-norm clipping (reference step: main.py:106-108, optimiser main.py:275).

``FlatParams`` re-homes a module's parameters and gradients into two contiguous f32 buffers
(device-agnostic, pure bookkeeping) so that (a) the optimiser is two kernels -- a sum-of-squares
reduction and one AdamW sweep that reads the clip coefficient on device, no host sync -- and
(b) data-parallel gradient exchange works on contiguous slices (``ddp.FlatDDP``).
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, List, Optional

import torch


class FlatParams:
    """Parameters and their gradients as views into two flat f32 buffers (registration order)."""

    def __init__(self, params: Iterable[torch.nn.Parameter]):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatParams: no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        if dt != torch.float32 or any(p.device != dev or p.dtype != dt for p in self.params):
            raise ValueError("FlatParams: all parameters must be float32 on one device")
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += p.numel()
        self.numel = off
        self.flat_p = torch.empty(off, dtype=dt, device=dev)
        self.flat_g = torch.zeros(off, dtype=dt, device=dev)
        for p, o in zip(self.params, self.offsets):
            self.flat_p[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.flat_p[o:o + p.numel()].view(p.shape)
        self.attach_grads()

    def attach_grads(self) -> None:
        """(Re)point every ``.grad`` at its slice of the flat buffer (autograd then accumulates in place)."""
        for p, o in zip(self.params, self.offsets):
            g = self.flat_g[o:o + p.numel()].view(p.shape)
            if p.grad is None or p.grad.data_ptr() != g.data_ptr():
                p.grad = g

    def zero_grad(self) -> None:
        self.flat_g.zero_()
        self.attach_grads()


class FusedAdamW(torch.optim.Optimizer):
    """AdamW(lr, betas, eps, weight_decay) + optional ``clip_grad_norm_(max_grad_norm)`` in two HIP kernels.

    Semantics equal ``torch.nn.utils.clip_grad_norm_(params, max_grad_norm)`` followed by
    ``torch.optim.AdamW.step()`` (main.py:106-108).  ``zero_grad`` keeps gradients as views of the
    flat buffer (``set_to_none`` is accepted and ignored).
    """

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm: Optional[float] = None,
                 loss_scale: Optional[float] = None, scale_growth: float = 2.0, scale_backoff: float = 0.5, scale_interval: int = 2000,
                 capturable: bool = False):
        params = list(params)
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        if len(self.param_groups) != 1:
            raise ValueError("FusedAdamW supports a single parameter group")
        self.flat = FlatParams(self.param_groups[0]["params"])
        if not self.flat.flat_p.is_cuda:
            raise RuntimeError("FusedAdamW needs HIP device parameters (no CPU path)")
        self.m = torch.zeros_like(self.flat.flat_p)
        self.v = torch.zeros_like(self.flat.flat_p)
        self.sumsq = torch.zeros(1, dtype=torch.float64, device=self.flat.flat_p.device)
        self.max_grad_norm = max_grad_norm
        self.step_count = 0
        # fp16 compute (ops.compute_dtype(torch.float16)): dynamic loss scaling, all on the device.  ``scale_state`` =
        # [scale, growth tracker, successful steps]; the training step multiplies the loss by scale_state[0] before backward.
        self.scale_state = None
        self._scale_cfg = (float(scale_growth), float(scale_backoff), int(scale_interval))
        if loss_scale is not None:
            self.scale_state = torch.tensor([float(loss_scale), 0.0, 0.0], dtype=torch.float32, device=self.flat.flat_p.device)
        # capturable: hyper-parameters and the step count live on the device (uclstm_adamw_step_dev), so that step() makes the
        # same launches with the same arguments every time and can be captured in a HIP graph (engine.GraphedTrainStep); a
        # changed lr / max_grad_norm reaches the device through one small copy in sync_hyper(), outside the graph.
        self.capturable = bool(capturable)
        self.hyper = None
        if self.capturable:
            if loss_scale is not None:
                raise ValueError("FusedAdamW(capturable=True) does not combine with loss scaling yet")
            self.hyper = torch.zeros(8, dtype=torch.float32, device=self.flat.flat_p.device)
            self._hyper_host = None
            self.sync_hyper()

    def _hyper_values(self):
        g = self.param_groups[0]
        return (float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]), float(g["weight_decay"]),
                float(self.max_grad_norm or 0.0))

    def sync_hyper(self) -> None:
        """Push lr / betas / eps / weight_decay / max_grad_norm to the device if they changed (capturable mode; never inside a
        capture).  The device-side step count is left alone."""
        if not self.capturable:
            return
        vals = self._hyper_values()
        if vals != self._hyper_host:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError("FusedAdamW.sync_hyper() inside a graph capture: hyper-parameters must be synchronised before")
            self.hyper[:6].copy_(torch.tensor(vals, dtype=torch.float32))
            self._hyper_host = vals

    def zero_grad(self, set_to_none: bool = False) -> None:   # noqa: ARG002 - signature parity with torch
        self.flat.zero_grad()

    # The moments and the bias-correction step live in flat buffers outside ``Optimizer.state``: carry them explicitly,
    # otherwise a save / load / resume would silently restart Adam.
    def state_dict(self):
        sd = super().state_dict()
        if self.capturable:
            self.step_count = int(self.hyper[6].item())          # the device-side count is the truth in capturable mode
        sd["fused"] = {"exp_avg": self.m.detach().clone(), "exp_avg_sq": self.v.detach().clone(), "step": int(self.step_count),
                       "numel": int(self.flat.numel),
                       "scale_state": None if self.scale_state is None else self.scale_state.detach().clone()}
        return sd

    def load_state_dict(self, state_dict) -> None:
        fused = state_dict.get("fused")
        if fused is None:
            raise ValueError("FusedAdamW.load_state_dict: no 'fused' entry (not a FusedAdamW state_dict)")
        if int(fused["numel"]) != self.flat.numel:
            raise ValueError(f"FusedAdamW.load_state_dict: {fused['numel']} parameters saved, {self.flat.numel} here")
        super().load_state_dict({k: v for k, v in state_dict.items() if k != "fused"})
        self.m.copy_(fused["exp_avg"])
        self.v.copy_(fused["exp_avg_sq"])
        self.step_count = int(fused["step"])
        if self.capturable:
            self.hyper[6] = float(self.step_count)
        if fused.get("scale_state") is not None and self.scale_state is not None:
            self.scale_state.copy_(fused["scale_state"])

    @torch.no_grad()
    def grad_norm(self) -> torch.Tensor:
        """Global L2 norm of the last step's (unscaled) gradients (device scalar, f64)."""
        n = self.sumsq.sqrt()
        return n if self.scale_state is None else n / self._last_scale.double()

    def scale_loss(self, loss: torch.Tensor) -> torch.Tensor:
        """The tensor to call ``backward()`` on: ``loss`` itself, or ``loss * scale`` with fp16 loss scaling."""
        return loss if self.scale_state is None else loss * self.scale_state[0]

    @torch.no_grad()
    def step(self, closure=None):
        from . import _lib as L
        from .ops import _stream
        assert closure is None
        g = self.param_groups[0]
        f = self.flat
        f.attach_grads()
        self.step_count += 1
        if self.scale_state is not None:
            self.sumsq.zero_()
            self._last_scale = self.scale_state[0].clone()
            L.check(L.lib.uclstm_sumsq(C.c_void_p(f.flat_g.data_ptr()), f.numel, C.c_void_p(self.sumsq.data_ptr()), _stream()), "sumsq")
            L.check(L.lib.uclstm_adamw_step_scaled(C.c_void_p(f.flat_p.data_ptr()), C.c_void_p(self.m.data_ptr()), C.c_void_p(self.v.data_ptr()),
                                                   C.c_void_p(f.flat_g.data_ptr()), f.numel, C.c_void_p(self.sumsq.data_ptr()),
                                                   float(self.max_grad_norm or 0.0), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                                                   float(g["eps"]), float(g["weight_decay"]), C.c_void_p(self.scale_state.data_ptr()), _stream()),
                    "adamw_step_scaled")
            gr, bo, it = self._scale_cfg
            L.check(L.lib.uclstm_loss_scale_update(C.c_void_p(self.scale_state.data_ptr()), C.c_void_p(self.sumsq.data_ptr()), gr, bo, it,
                                                   _stream()), "loss_scale_update")
            from . import ops
            ops.weights_changed()
            return None
        if self.capturable:
            if not torch.cuda.is_current_stream_capturing():
                self.sync_hyper()
            self.sumsq.zero_()
            L.check(L.lib.uclstm_sumsq(C.c_void_p(f.flat_g.data_ptr()), f.numel, C.c_void_p(self.sumsq.data_ptr()), _stream()), "sumsq")
            L.check(L.lib.uclstm_adamw_step_dev(C.c_void_p(f.flat_p.data_ptr()), C.c_void_p(self.m.data_ptr()), C.c_void_p(self.v.data_ptr()),
                                                C.c_void_p(f.flat_g.data_ptr()), f.numel, C.c_void_p(self.sumsq.data_ptr()),
                                                C.c_void_p(self.hyper.data_ptr()), _stream()), "adamw_step_dev")
            from . import ops
            ops.weights_changed()
            return None
        sq = None
        if self.max_grad_norm is not None:
            self.sumsq.zero_()
            L.check(L.lib.uclstm_sumsq(C.c_void_p(f.flat_g.data_ptr()), f.numel, C.c_void_p(self.sumsq.data_ptr()), _stream()), "sumsq")
            sq = C.c_void_p(self.sumsq.data_ptr())
        L.check(L.lib.uclstm_adamw_step(C.c_void_p(f.flat_p.data_ptr()), C.c_void_p(self.m.data_ptr()), C.c_void_p(self.v.data_ptr()),
                                        C.c_void_p(f.flat_g.data_ptr()), f.numel, sq,
                                        float(self.max_grad_norm or 0.0), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                                        float(g["eps"]), float(g["weight_decay"]), self.step_count, _stream()), "adamw_step")
        from . import ops
        ops.weights_changed()          # raw-pointer update: tensor versions do not move, panel caches must be told
        return None
