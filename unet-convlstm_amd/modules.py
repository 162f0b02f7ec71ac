"""nn.Module surface of the reference's train/unet.py, running on libuclstm.so.

Same class names, constructor signatures, attribute names and ``state_dict`` keys as the
reference, so checkpoints move both ways and the reference's training
loop drives these modules unchanged.  Parameters live in stock ``nn.Conv2d`` /
``nn.BatchNorm2d`` / ``nn.ConvTranspose2d`` containers (identical initialisation and RNG
consumption as the reference), but those containers are never *called*: every forward goes
through the HIP operators in ``ops.py``.

Public ``forward`` methods take/return the reference's tensors (f32 NCHW); the ``*_nhwc``
methods are the internal bf16 NHWC path that ``TemporalUNetDualView`` chains end to end with
all T timesteps batched (BatchNorm statistics stay per timestep).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from . import ops
from ._lib import UclstmError

Tensor = torch.Tensor


def _need_grad(*ts) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in ts)


# ---------------------------------------------------------------------------------------------
# ConvLSTM (reference train/unet.py:14-60)
# ---------------------------------------------------------------------------------------------
class ConvLSTMCell(nn.Module):
    """Reference train/unet.py:14-36.  ``cell(x, state=None) -> (h, (h, c))`` on f32 NCHW."""

    def __init__(self, input_dim, hidden_dim, kernel_size=3, bias=True):
        super().__init__()
        if kernel_size % 2 == 0 or not 1 <= kernel_size <= 7:
            # an even kernel with padding k//2 grows the map by one pixel per step: the reference's own cell update
            # (train/unet.py:34) then fails on shapes; the GEMM staging holds tap masks for up to 7x7 = 49 taps
            raise UclstmError("ConvLSTMCell: kernel_size must be odd and <= 7")
        padding = kernel_size // 2
        self.input_dim = input_dim
        self.hidden_dim = hidden_dim
        self.conv = nn.Conv2d(input_dim + hidden_dim, 4 * hidden_dim, kernel_size, padding=padding, bias=bias)

    # internal: whole sequence, NHWC
    def seq_nhwc(self, x_all: Tensor, h0: Optional[Tensor], c0: Optional[Tensor], out=None) -> Tuple[Tensor, Tensor]:
        """``out=(h_out, c_out)``: one-step inference writes the new state into these buffers (streaming.StreamingPredictor)."""
        need = _need_grad(x_all, h0, c0, self.conv.weight)
        if out is not None:
            return ops.ConvLSTMSeq.apply(x_all, h0, c0, self.conv.weight, self.conv.bias, self.hidden_dim, self.input_dim, need, out)
        return ops.ConvLSTMSeq.apply(x_all, h0, c0, self.conv.weight, self.conv.bias, self.hidden_dim, self.input_dim, need)

    def forward(self, x, state=None):
        B, Cc, H, W = x.shape
        xa = ops.ToNHWC.apply(x.contiguous().float()).unsqueeze(0)
        h0 = c0 = None
        if state is not None:
            h, c = state
            h0 = ops.ToNHWC.apply(h.contiguous().float())
            c0 = ops.StateToNHWC.apply(c.contiguous().float())
        h_all, c_T = self.seq_nhwc(xa, h0, c0)
        h_next = ops.FromNHWC.apply(h_all[0], self.hidden_dim)
        c_next = ops.StateFromNHWC.apply(c_T, self.hidden_dim)
        return h_next, (h_next, c_next)


class ConvLSTM(nn.Module):
    """Reference train/unet.py:39-60: layer-major, time-minor stack; ``x_seq`` is any indexable of T tensors."""

    def __init__(self, input_dim, hidden_dim, num_layers=1, kernel_size=3):
        super().__init__()
        self.layers = nn.ModuleList()
        for l in range(num_layers):
            self.layers.append(ConvLSTMCell(input_dim if l == 0 else hidden_dim, hidden_dim, kernel_size))

    def seq_nhwc(self, x_all: Tensor, state: Optional[Sequence], out_state: Optional[Sequence] = None):
        """x_all bf16 [T,B,H,W,Cp]; state: per layer None or (h bf16 NHWC, c f32 NHWC); out_state: per layer (h, c) buffers
        that receive the new state (one-step inference only)."""
        if state is None:
            state = [None] * len(self.layers)
        out = x_all
        new_states = []
        for li, layer in enumerate(self.layers):
            h0, c0 = (None, None) if state[li] is None else state[li]
            out, c_T = layer.seq_nhwc(out, h0, c0, None if out_state is None else out_state[li])
            new_states.append((out[-1], c_T))
        return out, new_states

    def forward(self, x_seq, state=None):
        T = len(x_seq)
        xs = torch.stack([ops.ToNHWC.apply(x_seq[t].contiguous().float()) for t in range(T)], dim=0)
        st = None
        if state is not None:
            st = []
            for s in state:
                if s is None or s[0] is None:
                    st.append(None)
                else:
                    st.append((ops.ToNHWC.apply(s[0].contiguous().float()), ops.StateToNHWC.apply(s[1].contiguous().float())))
        out, new_states = self.seq_nhwc(xs, st)
        hd = self.layers[-1].hidden_dim
        seq_out = [ops.FromNHWC.apply(out[t], hd) for t in range(T)]
        states = [(ops.FromNHWC.apply(h, l.hidden_dim), ops.StateFromNHWC.apply(c, l.hidden_dim))
                  for (h, c), l in zip(new_states, self.layers)]
        return seq_out, states


# ---------------------------------------------------------------------------------------------
# UNet blocks (reference train/unet.py:66-107)
# ---------------------------------------------------------------------------------------------
# BatchNorm's ``num_batches_tracked`` counters: 18 scalar int64 adds per model forward, 5 us each on the GPU.  The full model
# collects them here and bumps them with one multi-tensor add at the end of its forward (stand-alone blocks add directly).
_DEFERRED_COUNTERS: Optional[list] = None


def _flush_counters(pending: list) -> None:
    by_inc: dict = {}
    for t, inc in pending:
        by_inc.setdefault(inc, []).append(t)
    with torch.no_grad():
        for inc, ts in by_inc.items():
            torch._foreach_add_(ts, inc)


class DoubleConv(nn.Module):
    """Reference train/unet.py:66-75: (conv3x3 + BN + ReLU) x 2, ``self.net`` indices 0,1,3,4 hold the parameters."""

    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.net = nn.Sequential(
            nn.Conv2d(in_ch, out_ch, 3, padding=1), nn.BatchNorm2d(out_ch), nn.ReLU(inplace=True),
            nn.Conv2d(out_ch, out_ch, 3, padding=1), nn.BatchNorm2d(out_ch), nn.ReLU(inplace=True)
        )

    def _stage(self, conv: nn.Conv2d, bn: nn.BatchNorm2d, x0, x1, c_valid, off, groups, im2col=False, head=None, pool=False):
        """``head``: an ``nn.Conv2d(C, 1, 1)`` fused behind this stage (training mode): returns its f32 NCHW output.
        ``pool``: returns ``(activation, MaxPool2d(2)(activation))``."""
        training = self.training or not bn.track_running_stats
        if bn.momentum is not None:
            mom = bn.momentum
        elif training and bn.num_batches_tracked is not None:
            # momentum=None is PyTorch's cumulative moving average (factor 1/num_batches_tracked after the bump); the counter
            # lives on the device, so this rare mode (no reference script uses it) costs one host read per stage
            mom = -float(int(bn.num_batches_tracked) + 1)
        else:
            mom = 0.0
        if head is not None:
            a = ops.ConvBNReLU.apply(x0, x1, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                     tuple(c_valid), tuple(off), groups, training, mom, bn.eps, im2col, head.weight, head.bias)
        elif pool:
            a = ops.ConvBNReLU.apply(x0, x1, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                     tuple(c_valid), tuple(off), groups, training, mom, bn.eps, im2col, None, None, True)
        else:
            a = ops.ConvBNReLU.apply(x0, x1, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var,
                                     tuple(c_valid), tuple(off), groups, training, mom, bn.eps, im2col)
        if training and bn.num_batches_tracked is not None:
            if _DEFERRED_COUNTERS is not None:
                _DEFERRED_COUNTERS.append((bn.num_batches_tracked, groups))      # one multi-tensor add per model forward
            else:
                bn.num_batches_tracked += groups      # the reference calls BN once per timestep
        return a

    def forward_nhwc(self, x0: Tensor, x1: Optional[Tensor] = None, c_valid=None, off=(0, 0), groups: int = 1,
                     im2col: bool = False, head=None, pool: bool = False):
        conv0, bn0, conv1, bn1 = self.net[0], self.net[1], self.net[3], self.net[4]
        if c_valid is None:
            c_valid = (conv0.in_channels,)
        a = self._stage(conv0, bn0, x0, x1, c_valid, off, groups, im2col)
        return self._stage(conv1, bn1, a, None, (conv0.out_channels,), (0, 0), groups, head=head, pool=pool)

    def first_layer_nhwc(self, x: Tensor, time_major: bool, groups: int, pool: bool = False):
        """f32 NCHW (or [B,T,C,H,W] with ``time_major``) input that needs no gradient -> pre-gathered first conv."""
        cin = self.net[0].in_channels
        if 9 * cin <= 64 and not x.requires_grad:
            return self.forward_nhwc(ops.im2col_first(x.contiguous().float(), time_major), None, (cin,), (0, 0), groups, im2col=True,
                                     pool=pool)
        if time_major:
            B, T = x.shape[0], x.shape[1]
            x = x.transpose(0, 1).reshape(B * T, *x.shape[2:])
        return self.forward_nhwc(ops.ToNHWC.apply(x.contiguous().float()), None, (cin,), (0, 0), groups, pool=pool)

    def forward(self, x):
        a = self.first_layer_nhwc(x, False, 1)
        ops.join_forward_side(a.device)
        return ops.FromNHWC.apply(a, self.net[3].out_channels)


class Down(nn.Module):
    """Reference train/unet.py:78-84: ``Sequential(MaxPool2d(2), DoubleConv)`` -> keys ``net.1.net.N.*``."""

    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.net = nn.Sequential(nn.MaxPool2d(2), DoubleConv(in_ch, out_ch))

    def forward_nhwc(self, a: Tensor, groups: int = 1) -> Tensor:
        return self.net[1].forward_nhwc(ops.MaxPool2.apply(a), groups=groups)

    def forward_nhwc_skip(self, a: Tensor, groups: int = 1):
        """(block output, ``a`` for the skip connection): the caller must use the returned alias of ``a`` instead of ``a``, so
        that the two gradients of ``a`` meet in one backward kernel (ops.MaxPool2Skip)."""
        if not ops.POOL_SKIP:
            return self.forward_nhwc(a, groups), a
        p, skip = ops.MaxPool2Skip.apply(a)
        return self.net[1].forward_nhwc(p, groups=groups), skip

    def forward(self, x):
        a = self.forward_nhwc(ops.ToNHWC.apply(x.contiguous().float()))
        ops.join_forward_side(a.device)
        return ops.FromNHWC.apply(a, self.net[1].net[3].out_channels)


class Up(nn.Module):
    """Reference train/unet.py:87-98: ConvTranspose2d(k2,s2), centre-pad to the skip, cat([skip, up]), DoubleConv."""

    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.up = nn.ConvTranspose2d(in_ch, in_ch // 2, 2, stride=2)
        self.conv = DoubleConv(in_ch, out_ch)

    def forward_nhwc(self, x1: Tensor, x2: Tensor, skip_ch: int, groups: int = 1, head=None) -> Tensor:
        u = ops.ConvT2x2.apply(x1, self.up.weight, self.up.bias)
        diffY = x2.shape[1] - u.shape[1]
        diffX = x2.shape[2] - u.shape[2]
        # cat order is skip first (train/unet.py:98); a non-negative pad is folded into the source view offsets.  A NEGATIVE
        # difference (the upsampled map is larger than the skip: cannot happen inside the model, floor pooling only shrinks) is
        # F.pad's crop (train/unet.py:95-97; Python's floor division decides which side loses the odd pixel): the overhang is cut
        # off here with a slice copy -- the GEMM kernels place sources INSIDE the output frame only.
        if diffY < 0 or diffX < 0:
            t, l = max(-(diffY // 2), 0), max(-(diffX // 2), 0)
            hh = min(u.shape[1] - t, x2.shape[1]) if diffY < 0 else u.shape[1]
            ww = min(u.shape[2] - l, x2.shape[2]) if diffX < 0 else u.shape[2]
            u = u[:, t:t + hh, l:l + ww, :].contiguous()
            diffY, diffX = max(diffY, 0), max(diffX, 0)
        return self.conv.forward_nhwc(x2, u, (skip_ch, self.up.out_channels), (diffY // 2, diffX // 2), groups, head=head)

    def forward(self, x1, x2):
        a = self.forward_nhwc(ops.ToNHWC.apply(x1.contiguous().float()), ops.ToNHWC.apply(x2.contiguous().float()), x2.shape[1])
        ops.join_forward_side(a.device)
        return ops.FromNHWC.apply(a, self.conv.net[3].out_channels)


class OutConv(nn.Module):
    """Reference train/unet.py:101-107."""

    def __init__(self, in_ch, out_ch):
        super().__init__()
        self.conv = nn.Conv2d(in_ch, out_ch, 1)

    def forward_nhwc(self, a: Tensor) -> Tensor:
        return ops.OutConv1x1.apply(a, self.conv.weight, self.conv.bias)      # the parameter itself ([Co, Ci, 1, 1]): its .grad is written directly

    def forward(self, x):
        return self.forward_nhwc(ops.ToNHWC.apply(x.contiguous().float()))


class SpatialAttention(nn.Module):
    """Reference train/unet.py:113-125: channel mean & max -> k x k conv (2 -> 1, no bias) -> sigmoid -> scale, as HIP kernels
    (``uclstm_attention_fwd/bwd``).  ``self.conv`` / ``self.sigmoid`` hold the parameters under the reference's names
    (``attention.conv.weight``) and are never called."""

    def __init__(self, kernel_size=7):
        super().__init__()
        padding = kernel_size // 2
        self.conv = nn.Conv2d(2, 1, kernel_size, padding=padding, bias=False)
        self.sigmoid = nn.Sigmoid()

    def forward_nhwc(self, a: Tensor, channels: int) -> Tensor:
        return ops.SpatialAttn.apply(a, self.conv.weight, channels)

    def forward(self, x):
        a = self.forward_nhwc(ops.ToNHWC.apply(x.contiguous().float()), x.shape[1])
        return ops.FromNHWC.apply(a, x.shape[1])


# ---------------------------------------------------------------------------------------------
# TemporalUNetDualView (reference train/unet.py:131-204)
# ---------------------------------------------------------------------------------------------
class SeqList(list):
    """``list`` of per-timestep outputs with an optional ``stacked`` attribute: the same frames as one [B,T,...] view."""
    stacked: Optional[Tensor] = None


class TemporalUNetDualView(nn.Module):
    """Reference train/unet.py:131-204.

    ``model(x_seq[B,T,2*in_channels_per_sat,H,W], state=None) -> (list of T [B,out,H,W] f32, new_state)``
    where ``new_state`` is ``[(h, c)]`` per layer of ``self.temporal`` only (skip-LSTM states are
    dropped exactly like the reference, ``:190-191``).
    """

    def __init__(self, in_channels_per_sat=1, out_channels=1, base_ch=32, lstm_layers=1, use_skip_lstm=False, use_attention=False):
        super().__init__()
        in_ch_total = in_channels_per_sat * 2
        self.inc = DoubleConv(in_ch_total, base_ch)
        self.down1 = Down(base_ch, base_ch * 2)
        self.down2 = Down(base_ch * 2, base_ch * 4)
        self.down3 = Down(base_ch * 4, base_ch * 8)
        self.bottleneck = Down(base_ch * 8, base_ch * 16)

        self.use_attention = use_attention
        if self.use_attention:
            self.attention = SpatialAttention()

        self.temporal = ConvLSTM(base_ch * 16, base_ch * 16, num_layers=lstm_layers)

        self.use_skip_lstm = use_skip_lstm
        if use_skip_lstm:
            self.lstm_skip3 = ConvLSTM(base_ch * 8, base_ch * 8)
            self.lstm_skip2 = ConvLSTM(base_ch * 4, base_ch * 4)

        self.up3 = Up(base_ch * 16, base_ch * 8)
        self.up2 = Up(base_ch * 8, base_ch * 4)
        self.up1 = Up(base_ch * 4, base_ch * 2)
        self.up0 = Up(base_ch * 2, base_ch)
        self.outc = OutConv(base_ch, out_channels)
        self.base_ch = base_ch
        self.out_channels = out_channels

    # -- internal NHWC encoder over n images in `groups` BatchNorm groups
    def _encode_nhwc(self, x: Tensor, time_major: bool, groups: int):
        if ops.FUSE_POOL and ops.POOL_SKIP:
            # every block's second BatchNorm stage also writes its pooled output (and takes both gradients back in one kernel)
            x0, p0 = self.inc.first_layer_nhwc(x, time_major, groups, pool=True)
            x1, p1 = self.down1.net[1].forward_nhwc(p0, groups=groups, pool=True)
            x2, p2 = self.down2.net[1].forward_nhwc(p1, groups=groups, pool=True)
            x3, p3 = self.down3.net[1].forward_nhwc(p2, groups=groups, pool=True)
            xb = self.bottleneck.net[1].forward_nhwc(p3, groups=groups)
        else:
            x0 = self.inc.first_layer_nhwc(x, time_major, groups)
            x1, x0 = self.down1.forward_nhwc_skip(x0, groups)
            x2, x1 = self.down2.forward_nhwc_skip(x1, groups)
            x3, x2 = self.down3.forward_nhwc_skip(x2, groups)
            xb, x3 = self.bottleneck.forward_nhwc_skip(x3, groups)
        if self.use_attention:
            xb = self.attention.forward_nhwc(xb, self.base_ch * 16)
        return xb, (x3, x2, x1, x0)

    def _head_fusable(self) -> bool:
        """Training-mode forward that will be differentiated, ONE output channel, the last stage's channel chunks a power of two:
        the conditions of ops.ConvBNReLU's fused output head (UCLSTM_FUSE_HEAD=0 switches it off)."""
        bn = self.up0.conv.net[4]
        cpc = ops.cpad(self.base_ch) // 8
        return (ops.FUSE_HEAD and self.out_channels == 1 and (self.training or not bn.track_running_stats) and torch.is_grad_enabled()
                and cpc <= 64 and (cpc & (cpc - 1)) == 0)

    def encode_once(self, x_t):
        """Reference train/unet.py:161-172 on f32 NCHW (public helper, one timestep)."""
        xb, (x3, x2, x1, x0) = self._encode_nhwc(x_t, False, 1)
        ops.join_forward_side(xb.device)
        c = self.base_ch
        f = ops.FromNHWC.apply
        return f(xb, c * 16), (f(x3, c * 8), f(x2, c * 4), f(x1, c * 2), f(x0, c))

    def forward(self, x_seq, state=None):
        global _DEFERRED_COUNTERS
        pending, _DEFERRED_COUNTERS = [], None
        _DEFERRED_COUNTERS = pending
        try:
            return self._forward(x_seq, state)
        finally:
            _DEFERRED_COUNTERS = None
            if pending:
                _flush_counters(pending)
            if x_seq.is_cuda:
                ops.join_forward_side(x_seq.device)      # BatchNorm running statistics were updated on the second stream

    def _forward(self, x_seq, state=None):
        B, T, Cc, H, W = x_seq.shape
        c = self.base_ch
        # encoder: all T timesteps as one batch of T*B images (time-major), BN statistics per timestep
        xb, (x3, x2, x1, x0) = self._encode_nhwc(x_seq, True, T)

        def seq(t: Tensor) -> Tensor:          # [T*B,h,w,C] -> [T,B,h,w,C]
            return t.view(T, B, *t.shape[1:])

        st = None
        if state is not None:
            st = []
            for s in state:
                if s is None or s[0] is None:
                    st.append(None)
                else:
                    st.append((ops.ToNHWC.apply(s[0].contiguous().float()), ops.StateToNHWC.apply(s[1].contiguous().float())))
        grouped = False
        if self.use_skip_lstm and len(self.temporal.layers) == 1:
            # the three recurrences are independent (train/unet.py:185-191 runs them one after the other): one group launch
            # per timestep when every member's step GEMM takes the patch shape, else three sequences of launches
            cells = (self.temporal.layers[0], self.lstm_skip3.layers[0], self.lstm_skip2.layers[0])
            h0, c0 = (None, None) if (st is None or st[0] is None) else st[0]
            members = [(seq(xb), h0, c0), (seq(x3), None, None), (seq(x2), None, None)]
            full = [(xs, hh, cc, cl.conv.weight, cl.conv.bias, cl.hidden_dim, cl.input_dim) for (xs, hh, cc), cl in zip(members, cells)]
            if ops.convlstm_group_ok(full):
                needs = [_need_grad(*f[:5]) for f in full]
                pre = ops.convlstm_group_forward(full, any(needs))
                outs = [ops.ConvLSTMSeqPre.apply(*f, nd, hh, ch, gt) for f, nd, (hh, ch, gt) in zip(full, needs, pre)]
                b_all, new_st = outs[0][0], [(outs[0][0][-1], outs[0][1])]
                x3_l, x2_l = outs[1][0], outs[2][0]
                grouped = True
        if not grouped:
            b_all, new_st = self.temporal.seq_nhwc(seq(xb), st)
            if self.use_skip_lstm:
                x3_l, _ = self.lstm_skip3.seq_nhwc(seq(x3), None)
                x2_l, _ = self.lstm_skip2.seq_nhwc(seq(x2), None)
        if self.use_skip_lstm:
            x3 = x3_l.reshape(T * B, *x3_l.shape[2:])
            x2 = x2_l.reshape(T * B, *x2_l.shape[2:])
        b_flat = b_all.reshape(T * B, *b_all.shape[2:])

        d3 = self.up3.forward_nhwc(b_flat, x3, c * 8, T)
        d2 = self.up2.forward_nhwc(d3, x2, c * 4, T)
        d1 = self.up1.forward_nhwc(d2, x1, c * 2, T)
        if self._head_fusable():
            # up0's second stage and the 1x1 output convolution as one op: that activation and its gradient never exist in memory
            y = self.up0.forward_nhwc(d1, x0, c, T, head=self.outc.conv).view(T, B, self.out_channels, H, W)
        else:
            d0 = self.up0.forward_nhwc(d1, x0, c, T)
            y = self.outc.forward_nhwc(d0).view(T, B, self.out_channels, H, W)
        # a plain list of T frames like the reference's (train/unet.py:200-203); it also carries the frames already laid out
        # as [B,T,C,H,W] (a view) so that the training loop's torch.stack(output, dim=1) costs no copy kernels and its backward
        # is one transpose instead of T zero-fill + accumulate pairs
        out_seq = SeqList(y.unbind(0))
        out_seq.stacked = y.transpose(0, 1)

        new_state = [(ops.FromNHWC.apply(h, c * 16), ops.StateFromNHWC.apply(cc, c * 16)) for (h, cc) in new_st]
        return out_seq, new_state


    # -- streaming: one frame in, one frame out, ALL recurrent states carried (API superset of the reference, whose
    #    forward() drops the skip-LSTM states, train/unet.py:190-191)
    def step_nhwc(self, x_t: Tensor, full_state: Optional[dict] = None, out_state: Optional[dict] = None):
        """``x_t`` f32 ``[B, 2*in_channels_per_sat, H, W]`` -> ``(y_t f32 [B,out,H,W], full_state)``.

        ``full_state`` maps ``'temporal'/'skip3'/'skip2'`` to per-layer ``(h bf16 NHWC, c f32 NHWC)`` lists; ``None`` is the
        zero state.  Feeding frames one by one reproduces ``forward()`` on the whole sequence, skip LSTMs included.
        ``out_state`` (same structure, inference only): buffers the new state is written into -- no copies; h buffers must
        differ from the input state's, c buffers may be the same tensors."""
        c = self.base_ch
        st = full_state or {}
        ost = out_state or {}
        xb, (x3, x2, x1, x0) = self._encode_nhwc(x_t, False, 1)
        grouped = False
        if (self.use_skip_lstm and len(self.temporal.layers) == 1 and not torch.is_grad_enabled()
                and all(st.get(k) is not None and ost.get(k) is not None for k in ("temporal", "skip3", "skip2"))):
            # streaming inference with every state carried and caller-owned output buffers: the three cell steps as ONE launch
            cells = (self.temporal.layers[0], self.lstm_skip3.layers[0], self.lstm_skip2.layers[0])
            names, xs = ("temporal", "skip3", "skip2"), (xb, x3, x2)
            members = [(xx, st[k][0][0], st[k][0][1], ost[k][0][0], ost[k][0][1], cl.conv.weight, cl.conv.bias, cl.hidden_dim, cl.input_dim)
                       for k, xx, cl in zip(names, xs, cells)]
            if ops.convlstm_group_step(members):
                grouped = True
                b_all = ost["temporal"][0][0].unsqueeze(0)
                x3, x2 = ost["skip3"][0][0], ost["skip2"][0][0]
                new_state = {k: [(ost[k][0][0], ost[k][0][1])] for k in names}
        if not grouped:
            b_all, st_t = self.temporal.seq_nhwc(xb.unsqueeze(0), st.get("temporal"), ost.get("temporal"))
            new_state = {"temporal": st_t}
            if self.use_skip_lstm:
                x3_l, st3 = self.lstm_skip3.seq_nhwc(x3.unsqueeze(0), st.get("skip3"), ost.get("skip3"))
                x2_l, st2 = self.lstm_skip2.seq_nhwc(x2.unsqueeze(0), st.get("skip2"), ost.get("skip2"))
                x3, x2 = x3_l[0], x2_l[0]
                new_state["skip3"], new_state["skip2"] = st3, st2
        d3 = self.up3.forward_nhwc(b_all[0], x3, c * 8, 1)
        d2 = self.up2.forward_nhwc(d3, x2, c * 4, 1)
        d1 = self.up1.forward_nhwc(d2, x1, c * 2, 1)
        d0 = self.up0.forward_nhwc(d1, x0, c, 1)
        ops.join_forward_side(d0.device)
        return self.outc.forward_nhwc(d0), new_state


UNet = TemporalUNetDualView   # BASELINE.json calls the model "UNet"; the reference class is TemporalUNetDualView
