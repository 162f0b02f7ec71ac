"""Stateful frame-by-frame inference (SURVEY.md section 8f-4 / BASELINE.json config 5).

The reference's inference script re-runs the model on growing prefixes (test.py:305-310: O(T^2) frames, no carried
state).  ``StreamingPredictor`` keeps every recurrent state (temporal AND skip LSTMs) on the device and advances one
frame per call; with ``use_graph=True`` the whole step -- encoder, three ConvLSTM cell steps, decoder, state update --
is captured into a HIP graph (one per state parity: the hidden states alternate between two buffer sets, the cell
states are updated in place, so carrying the state costs no copies) and replayed: a step is one graph launch instead of
~70 kernel launches.
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
        self._states: Optional[list] = None          # two state sets: frame n reads [n & 1] and writes [1 - (n & 1)]
        self._parity = 0
        self._shape = None
        self._drop_graph_and_panels()

    def new_sequence(self) -> None:
        """Zero the recurrent state IN PLACE (the captured graphs keep updating these very buffers): the next frame starts a
        new sequence of the same shape, the graphs and the packed panels stay."""
        if self._states is not None:
            for st in self._states:
                for layers in st.values():
                    for h, c in layers:
                        h.zero_()
                        c.zero_()

    @property
    def state(self) -> Optional[dict]:
        """The current recurrent state: ``{'temporal' | 'skip3' | 'skip2': [(h NHWC, c f32 NHWC) per layer]}`` (live buffers)."""
        return None if self._states is None else self._states[self._parity]

    def _drop_graph_and_panels(self) -> None:
        # order matters: the captured graph has the panels' addresses baked in, so it goes first; the panels are owned by
        # this predictor's cache (never by a module-global one) and die with it
        self._graphs = [None, None]                   # one captured step per state parity
        self._static_x = None
        self._static_y = [None, None]
        self._eager_steps = 0
        self._panels = ops.PanelCache()
        self._wsig = self._weights_signature()

    def _weights_signature(self):
        return (ops.WEIGHTS_EPOCH, sum(p._version for p in self.model.parameters()) + sum(b._version for b in self.model.buffers()))

    # ---- state: fixed buffers so that captured graphs can update them in place, and no copies: the cell state c is updated
    #      in place (the cell update is element-wise), the hidden state h alternates between two buffers (the gate convolution
    #      reads h_{t-1}'s neighbourhoods while h_t is written).  Frames of even parity read set 0 and write set 1.
    def _zero_states(self, x_t: torch.Tensor) -> list:
        m, dev = self.model, x_t.device
        B, _, H, W = x_t.shape
        c = m.base_ch
        spec = {"temporal": (c * 16, H // 16, W // 16, len(m.temporal.layers))}
        if m.use_skip_lstm:
            spec["skip3"] = (c * 8, H // 8, W // 8, 1)
            spec["skip2"] = (c * 4, H // 4, W // 4, 1)
        sets = [{}, {}]
        for name, (ch, h, w, n_layers) in spec.items():
            cs = [torch.zeros((B, h, w, ops.cpad(ch)), dtype=torch.float32, device=dev) for _ in range(n_layers)]
            for st in sets:
                st[name] = [(torch.zeros((B, h, w, ops.cpad(ch)), dtype=ops.get_compute_dtype(), device=dev), cs[i]) for i in range(n_layers)]
        return sets

    def _eager(self, x_t: torch.Tensor, parity: int) -> torch.Tensor:
        y, _ = self.model.step_nhwc(x_t, self._states[parity], self._states[1 - parity])
        return y

    @torch.no_grad()
    def step(self, x_t: torch.Tensor) -> torch.Tensor:
        """One frame ``[B, C, H, W]`` (f32, device) -> prediction ``[B, out, H, W]`` (a fresh tensor)."""
        x_t = x_t.contiguous().float()
        if self._shape != tuple(x_t.shape):
            if self._shape is not None:
                raise ValueError("StreamingPredictor: frame shape changed; call reset() first")
            self._shape = tuple(x_t.shape)
            self._states = self._zero_states(x_t)
            self._parity = 0
        sig = self._weights_signature()
        if self._panels.stale() or sig != self._wsig:
            # the model was trained on (fused optimiser step, or any in-place update that bumped a tensor version) since the
            # panels were packed: keep the recurrent state, drop everything that holds old weights
            self._drop_graph_and_panels()
            self._wsig = sig
        p = self._parity
        self._parity = 1 - p
        with self._panels:                            # weights are frozen between optimiser steps: pack each panel once
            if not self.use_graph:
                return self._eager(x_t, p)
            if self._graphs[p] is None:
                if self._eager_steps < self.warmup:   # eager warm-up: first-launch attribute calls, panel cache fill
                    self._eager_steps += 1
                    return self._eager(x_t, p)
                if self._static_x is None:
                    self._static_x = x_t.clone()
                else:
                    self._static_x.copy_(x_t)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._static_y[p] = self._eager(self._static_x, p)
                self._graphs[p] = g
                # the capture only RECORDED the step for x_t: replay it now so that this frame is actually consumed
            self._static_x.copy_(x_t)
            self._graphs[p].replay()
            return self._static_y[p].clone()

    @torch.no_grad()
    def rollout(self, x_seq: torch.Tensor) -> torch.Tensor:
        """``[B, T, C, H, W]`` -> ``[B, T, out, H, W]`` frame by frame from the current state."""
        return torch.stack([self.step(x_seq[:, t]) for t in range(x_seq.shape[1])], dim=1)
