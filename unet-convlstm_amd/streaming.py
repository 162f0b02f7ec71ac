"""Stateful frame-by-frame inference (SURVEY.md section 8f-4 / BASELINE.json config 5).

The reference's inference script re-runs the model on growing prefixes (test.py:305-310: O(T^2) frames, no carried
state).  ``StreamingPredictor`` keeps every recurrent state (temporal AND skip LSTMs) on the device and advances one
frame per call; with ``use_graph=True`` the whole step -- encoder, three ConvLSTM cell steps, decoder, state update --
is captured once into a HIP graph and replayed, so a step costs one graph launch instead of ~70 kernel launches.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from .modules import TemporalUNetDualView


class StreamingPredictor:
    def __init__(self, model: TemporalUNetDualView, use_graph: bool = True, warmup: int = 2):
        self.model = model.eval()
        self.use_graph = use_graph
        self.warmup = warmup
        self.reset()

    def reset(self) -> None:
        """Forget the recurrent state (next frame starts a new sequence), any captured graph and the packed panels."""
        self._state: Optional[dict] = None
        self._shape = None
        self._drop_graph_and_panels()

    def new_sequence(self) -> None:
        """Zero the recurrent state IN PLACE (the captured graph keeps updating these very buffers): the next frame starts a
        new sequence of the same shape, the graph and the packed panels stay."""
        if self._state is not None:
            for layers in self._state.values():
                for h, c in layers:
                    h.zero_()
                    c.zero_()

    def _drop_graph_and_panels(self) -> None:
        # order matters: the captured graph has the panels' addresses baked in, so it goes first; the panels are owned by
        # this predictor's cache (never by a module-global one) and die with it
        self._graph = None
        self._static_x = None
        self._static_y = None
        self._eager_steps = 0
        self._panels = ops.PanelCache()
        self._wsig = self._weights_signature()

    def _weights_signature(self):
        return (ops.WEIGHTS_EPOCH, sum(p._version for p in self.model.parameters()) + sum(b._version for b in self.model.buffers()))

    # ---- state helpers: fixed buffers so that a captured graph can update them in place ----
    def _zero_state(self, x_t: torch.Tensor) -> dict:
        m, dev = self.model, x_t.device
        B, _, H, W = x_t.shape
        c = m.base_ch

        def z(ch, h, w, n_layers):
            return [(torch.zeros((B, h, w, ops.cpad(ch)), dtype=ops.get_compute_dtype(), device=dev),
                     torch.zeros((B, h, w, ops.cpad(ch)), dtype=torch.float32, device=dev)) for _ in range(n_layers)]
        st = {"temporal": z(c * 16, H // 16, W // 16, len(m.temporal.layers))}
        if m.use_skip_lstm:
            st["skip3"] = z(c * 8, H // 8, W // 8, 1)
            st["skip2"] = z(c * 4, H // 4, W // 4, 1)
        return st

    @staticmethod
    def _assign(dst: dict, src: dict) -> None:
        for k, layers in src.items():
            for (dh, dc), (sh, sc) in zip(dst[k], layers):
                dh.copy_(sh)
                dc.copy_(sc)

    def _eager(self, x_t: torch.Tensor) -> torch.Tensor:
        y, new_state = self.model.step_nhwc(x_t, self._state)
        self._assign(self._state, new_state)
        return y

    @torch.no_grad()
    def step(self, x_t: torch.Tensor) -> torch.Tensor:
        """One frame ``[B, C, H, W]`` (f32, device) -> prediction ``[B, out, H, W]`` (a fresh tensor)."""
        x_t = x_t.contiguous().float()
        if self._shape != tuple(x_t.shape):
            if self._shape is not None:
                raise ValueError("StreamingPredictor: frame shape changed; call reset() first")
            self._shape = tuple(x_t.shape)
            self._state = self._zero_state(x_t)
        sig = self._weights_signature()
        if self._panels.stale() or sig != self._wsig:
            # the model was trained on (fused optimiser step, or any in-place update that bumped a tensor version) since the
            # panels were packed: keep the recurrent state, drop everything that holds old weights
            self._drop_graph_and_panels()
            self._wsig = sig
        with self._panels:                            # weights are frozen between optimiser steps: pack each panel once
            if not self.use_graph:
                return self._eager(x_t)
            if self._graph is None:
                if self._eager_steps < self.warmup:   # eager warm-up: first-launch attribute calls, panel cache fill
                    self._eager_steps += 1
                    return self._eager(x_t)
                self._static_x = x_t.clone()
                self._graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self._graph):
                    self._static_y = self._eager(self._static_x)
                # the capture only RECORDED the step for x_t: replay it now so that this frame is actually consumed
            self._static_x.copy_(x_t)
            self._graph.replay()
            return self._static_y.clone()

    @torch.no_grad()
    def rollout(self, x_seq: torch.Tensor) -> torch.Tensor:
        """``[B, T, C, H, W]`` -> ``[B, T, out, H, W]`` frame by frame from the current state."""
        return torch.stack([self.step(x_seq[:, t]) for t in range(x_seq.shape[1])], dim=1)
