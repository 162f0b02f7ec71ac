"""fp32 CPU restatement of the reference's UNet-ConvLSTM training path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Every function states
the reference lines it follows (paths relative to the reference checkout).
The restatement is *functional*: parameters live in a flat ``dict`` keyed by the
reference's ``state_dict`` names, BatchNorm and the
LSTM gate algebra are written out as tensor arithmetic, and gradients come
from ``torch.autograd`` over these functions.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function
here against fixtures in ``tests/golden/`` that were produced by importing the
reference's own ``train/unet.py`` / ``main.py`` (generator:
``tests/golden/make_golden.py``).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
Params = Dict[str, Tensor]

BN_EPS = 1e-5        # nn.BatchNorm2d default used at train/unet.py:70-71
BN_MOMENTUM = 0.1    # idem


# ---------------------------------------------------------------------------
# Optional storage-precision emulation (checker-side tool, off by default)
# ---------------------------------------------------------------------------
# The HIP path stores activations and weight panels in bf16 and accumulates in
# f32.  Inside ``with bf16_storage():`` the oracle rounds to bf16 at exactly the
# points where the kernels store (conv/convT outputs, BN+ReLU outputs, LSTM h,
# weights, the network input) and is otherwise unchanged f32 arithmetic.  The
# rounding is straight-through for autograd, i.e. gradients are the exact f32
# gradients of a network that has the kernel's forward activations (and hence
# the kernel's ReLU masks).  This separates "is the kernel arithmetic right"
# (tight tolerance against this mode) from "how far does bf16 storage drift
# from the f32 reference" (stated tolerance against the plain oracle).
class _RoundSTE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return t.to(_STORAGE_DTYPE).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        # The storage points of the forward pass are the storage points of the backward pass too: the gradient of a stored
        # conv output is the kernels' dz, that of a stored activation their dx -- both 16-bit tensors on the HIP path.
        return g.to(_STORAGE_DTYPE).to(torch.float32) if _ROUND_GRADS else g


_ROUND_GRADS = False
_EMULATE = False
_STORAGE_DTYPE = torch.bfloat16
_EVAL_KEEPS_CONV_OUT = False


class storage:
    """Round to ``dtype`` (torch.bfloat16 / torch.float16) wherever the HIP path of that compute dtype stores.

    ``eval_with_backward``: an evaluation-mode BatchNorm that WILL be differentiated through (fine-tuning with frozen
    statistics) keeps its pre-BN convolution output in storage precision like the training path does (backward needs it),
    whereas pure inference folds BatchNorm into the convolution epilogue and rounds once.

    ``round_grads``: the gradients flowing back through the storage points are rounded as well (the kernels store dz and dx
    in 16 bits).  The mode must still be active when ``backward()`` runs."""

    def __init__(self, dtype, eval_with_backward: bool = False, round_grads: bool = False):
        self.dtype, self.eval_bw, self.round_grads = dtype, eval_with_backward, round_grads

    def __enter__(self):
        global _EMULATE, _STORAGE_DTYPE, _EVAL_KEEPS_CONV_OUT, _ROUND_GRADS
        self._old = (_EMULATE, _STORAGE_DTYPE, _EVAL_KEEPS_CONV_OUT, _ROUND_GRADS)
        _EMULATE, _STORAGE_DTYPE, _EVAL_KEEPS_CONV_OUT, _ROUND_GRADS = True, self.dtype, self.eval_bw, self.round_grads

    def __exit__(self, *exc):
        global _EMULATE, _STORAGE_DTYPE, _EVAL_KEEPS_CONV_OUT, _ROUND_GRADS
        _EMULATE, _STORAGE_DTYPE, _EVAL_KEEPS_CONV_OUT, _ROUND_GRADS = self._old


def bf16_storage(eval_with_backward: bool = False, round_grads: bool = False):
    return storage(torch.bfloat16, eval_with_backward, round_grads)


def fp16_storage():
    """The fp16-MFMA twin kernels (BASELINE.json configs[3]) store IEEE binary16."""
    return storage(torch.float16)


def _q(t: Tensor) -> Tensor:
    return _RoundSTE.apply(t) if _EMULATE else t


class _RoundWeight(torch.autograd.Function):
    """Weights are packed into 16-bit panels, their GRADIENTS stay f32 (f32 slabs -> f32 .grad)."""

    @staticmethod
    def forward(ctx, t):
        return t.to(_STORAGE_DTYPE).to(torch.float32)

    @staticmethod
    def backward(ctx, g):
        return g


def _qw(t: Tensor) -> Tensor:
    return _RoundWeight.apply(t) if _EMULATE else t


# ---------------------------------------------------------------------------
# ConvLSTM (train/unet.py:14-60)
# ---------------------------------------------------------------------------
def convlstm_cell(x: Tensor, h: Optional[Tensor], c: Optional[Tensor],
                  weight: Tensor, bias: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
    """One cell step, train/unet.py:21-36.

    ``weight`` is the single gate convolution ``[4*Hd, Cin+Hd, k, k]``
    (train/unet.py:19); input channels are ordered x then h (``:28``), output
    channels are four contiguous blocks i, f, g, o (``:29``).  ``h``/``c`` of
    ``None`` mean the zero state (``:23-25``).
    """
    hd = weight.shape[0] // 4
    k = weight.shape[-1]
    B, _, H, W = x.shape
    if h is None:
        h = x.new_zeros(B, hd, H, W)
        c = x.new_zeros(B, hd, H, W)
    pre = F.conv2d(torch.cat((_q(x), _q(h)), dim=1), _qw(weight), bias, padding=k // 2)
    pi, pf, pg, po = pre[:, 0:hd], pre[:, hd:2 * hd], pre[:, 2 * hd:3 * hd], pre[:, 3 * hd:4 * hd]
    gi = 1.0 / (1.0 + torch.exp(-pi))
    gf = 1.0 / (1.0 + torch.exp(-pf))
    gg = torch.tanh(pg)
    go = 1.0 / (1.0 + torch.exp(-po))
    c_new = gf * c + gi * gg            # :34
    h_new = _q(go * torch.tanh(c_new))  # :35
    return h_new, c_new


def convlstm(x_seq: Sequence[Tensor], p: Params, prefix: str, num_layers: int,
             state: Optional[List[Optional[Tuple[Tensor, Tensor]]]] = None):
    """Layer-major, time-minor stack, train/unet.py:46-60.

    Returns ``(list[T] of h of the last layer, list[(h, c)] per layer)``.
    """
    T = len(x_seq)
    if state is None:
        state = [None] * num_layers
    seq = list(x_seq)
    new_states = []
    for li in range(num_layers):
        w = p[f"{prefix}.layers.{li}.conv.weight"]
        b = p.get(f"{prefix}.layers.{li}.conv.bias")
        h, c = (None, None) if state[li] is None else state[li]
        outs = []
        for t in range(T):
            h, c = convlstm_cell(seq[t], h, c, w, b)
            outs.append(h)
        seq = outs
        new_states.append((h, c))
    return seq, new_states


# ---------------------------------------------------------------------------
# UNet blocks (train/unet.py:66-107)
# ---------------------------------------------------------------------------
def batchnorm_relu(z: Tensor, p: Params, prefix: str, training: bool,
                   buffers_out: Optional[Params]) -> Tensor:
    """BatchNorm2d(eps 1e-5, momentum 0.1, affine) + ReLU, train/unet.py:70-71.

    Training: normalise with the biased batch variance over (B,H,W) of THIS
    call, update running stats with the unbiased variance, bump
    ``num_batches_tracked`` (one call = one timestep, SURVEY.md section 7
    hard part 1).  Updated buffers are written to ``buffers_out``.
    """
    gamma, beta = p[f"{prefix}.weight"], p[f"{prefix}.bias"]
    if training:
        z = _q(z)         # the HIP path stores the conv output before normalising it
        n = z.shape[0] * z.shape[2] * z.shape[3]
        mean = z.mean(dim=(0, 2, 3))
        var = ((z - mean[None, :, None, None]) ** 2).mean(dim=(0, 2, 3))
        if buffers_out is not None:
            rm = buffers_out.get(f"{prefix}.running_mean", p[f"{prefix}.running_mean"])
            rv = buffers_out.get(f"{prefix}.running_var", p[f"{prefix}.running_var"])
            nb = buffers_out.get(f"{prefix}.num_batches_tracked", p[f"{prefix}.num_batches_tracked"])
            unbiased = var.detach() * (n / max(n - 1, 1))
            buffers_out[f"{prefix}.running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean.detach()
            buffers_out[f"{prefix}.running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * unbiased
            buffers_out[f"{prefix}.num_batches_tracked"] = nb + 1
    else:
        if _EVAL_KEEPS_CONV_OUT:
            z = _q(z)
        src = p if buffers_out is None else {**p, **buffers_out}
        mean, var = src[f"{prefix}.running_mean"], src[f"{prefix}.running_var"]
    xhat = (z - mean[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + BN_EPS)
    return _q(torch.clamp_min(xhat * gamma[None, :, None, None] + beta[None, :, None, None], 0.0))


def double_conv(x: Tensor, p: Params, prefix: str, training: bool,
                buffers_out: Optional[Params]) -> Tensor:
    """(conv3x3 pad1 + bias -> BN -> ReLU) x2, train/unet.py:66-75 (Sequential indices 0,1,3,4)."""
    z = F.conv2d(_q(x), _qw(p[f"{prefix}.net.0.weight"]), p[f"{prefix}.net.0.bias"], padding=1)
    a = batchnorm_relu(z, p, f"{prefix}.net.1", training, buffers_out)
    z = F.conv2d(a, _qw(p[f"{prefix}.net.3.weight"]), p[f"{prefix}.net.3.bias"], padding=1)
    return batchnorm_relu(z, p, f"{prefix}.net.4", training, buffers_out)


def down(x: Tensor, p: Params, prefix: str, training: bool, buffers_out) -> Tensor:
    """MaxPool2d(2) then DoubleConv, train/unet.py:78-84 (keys ``<prefix>.net.1.net.N``)."""
    # storage emulation: the HIP path pools the STORED (rounded) tensor -- same values as rounding after the pool, but ties of
    # rounded values route the gradient to the first maximum in scan order (ATen's rule, which the kernel reproduces)
    return double_conv(F.max_pool2d(_q(x), 2), p, f"{prefix}.net.1", training, buffers_out)


def up(x1: Tensor, x2: Tensor, p: Params, prefix: str, training: bool, buffers_out) -> Tensor:
    """ConvTranspose2d(k2,s2) -> pad to skip size -> cat([skip, up]) -> DoubleConv, train/unet.py:87-98."""
    u = _q(F.conv_transpose2d(_q(x1), _qw(p[f"{prefix}.up.weight"]), p[f"{prefix}.up.bias"], stride=2))
    dy = x2.shape[2] - u.shape[2]
    dx = x2.shape[3] - u.shape[3]
    u = F.pad(u, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])     # :95-97
    return double_conv(torch.cat((x2, u), dim=1), p, f"{prefix}.conv", training, buffers_out)


def out_conv(x: Tensor, p: Params, prefix: str) -> Tensor:
    """1x1 convolution, train/unet.py:101-107."""
    return F.conv2d(_q(x), p[f"{prefix}.conv.weight"], p[f"{prefix}.conv.bias"])


def spatial_attention(x: Tensor, p: Params, prefix: str) -> Tensor:
    """Channel mean & max -> 7x7 conv (2->1, no bias) -> sigmoid -> scale, train/unet.py:113-125."""
    w = p[f"{prefix}.conv.weight"]
    desc = torch.cat((x.mean(dim=1, keepdim=True), x.max(dim=1, keepdim=True).values), dim=1)
    att = torch.sigmoid(F.conv2d(desc, w, None, padding=w.shape[-1] // 2))
    return _q(x * att)          # the HIP path stores the scaled map in 16 bits (descriptor, conv and sigmoid stay f32)


# ---------------------------------------------------------------------------
# TemporalUNetDualView (train/unet.py:131-204)
# ---------------------------------------------------------------------------
def model_config_from_params(p: Params) -> dict:
    """Recover the constructor switches from the state_dict keys."""
    n_layers = 0
    while f"temporal.layers.{n_layers}.conv.weight" in p:
        n_layers += 1
    return {
        "base_ch": p["inc.net.0.weight"].shape[0],
        "lstm_layers": n_layers,
        "use_skip_lstm": "lstm_skip3.layers.0.conv.weight" in p,
        "use_attention": "attention.conv.weight" in p,
    }


def encode_once(x_t: Tensor, p: Params, training: bool, buffers_out, use_attention: bool):
    """train/unet.py:161-172."""
    x0 = double_conv(x_t, p, "inc", training, buffers_out)
    x1 = down(x0, p, "down1", training, buffers_out)
    x2 = down(x1, p, "down2", training, buffers_out)
    x3 = down(x2, p, "down3", training, buffers_out)
    xb = down(x3, p, "bottleneck", training, buffers_out)
    if use_attention:
        xb = spatial_attention(xb, p, "attention")
    return xb, (x3, x2, x1, x0)


def model_forward(p: Params, x_seq: Tensor, state=None, training: bool = True,
                  buffers_out: Optional[Params] = None):
    """Full forward, train/unet.py:174-204.

    ``x_seq`` is ``[B,T,2*in_channels_per_sat,H,W]``; returns the list of T
    ``[B,out,H,W]`` outputs and the new state of ``temporal`` only (the skip
    LSTM states are dropped, ``:190-191``).  BatchNorm runs once per timestep
    in encoder order t=0..T-1, then decoder order t=0..T-1 (``:179``, ``:196``).
    """
    cfg = model_config_from_params(p)
    T = x_seq.shape[1]
    bottlenecks, skips = [], []
    for t in range(T):
        xb, sk = encode_once(x_seq[:, t], p, training, buffers_out, cfg["use_attention"])
        bottlenecks.append(xb)
        skips.append(sk)
    b_out, new_state = convlstm(bottlenecks, p, "temporal", cfg["lstm_layers"], state)
    if cfg["use_skip_lstm"]:
        x3_l, _ = convlstm([s[0] for s in skips], p, "lstm_skip3", 1)
        x2_l, _ = convlstm([s[1] for s in skips], p, "lstm_skip2", 1)
        skips = [(x3_l[t], x2_l[t], skips[t][2], skips[t][3]) for t in range(T)]
    outs = []
    for t in range(T):
        x3, x2, x1, x0 = skips[t]
        d3 = up(b_out[t], x3, p, "up3", training, buffers_out)
        d2 = up(d3, x2, p, "up2", training, buffers_out)
        d1 = up(d2, x1, p, "up1", training, buffers_out)
        d0 = up(d1, x0, p, "up0", training, buffers_out)
        outs.append(out_conv(d0, p, "outc"))
    return outs, new_state


# ---------------------------------------------------------------------------
# Loss (main.py:28-72)
# ---------------------------------------------------------------------------
def compute_loss(y_pred: Tensor, y: Tensor, mask: Optional[Tensor] = None, use_mask: bool = True) -> Tensor:
    """Weighted L1 + 0.005 x spatial-gradient L1, main.py:28-72.

    weight = 1 + 4|y|^3 (``:38``); masked form divides by ``sum(mask*weight)+1e-8``
    (``:41-43``); forward differences along W and H cropped to (H-1, W-1)
    (``:48-62``).  Inputs are 5-D ``[B,T,1,H,W]`` (``:57-58``).
    """
    ad = (y_pred - y).abs()
    w = 1.0 + 4.0 * y.abs() ** 3
    masked = use_mask and mask is not None
    if masked:
        l1 = (ad * mask * w).sum() / ((mask * w).sum() + 1e-8)
    else:
        l1 = (ad * w).mean()
    H, W = y.shape[-2], y.shape[-1]
    dxp = y_pred[..., :, 1:] - y_pred[..., :, :-1]
    dyp = y_pred[..., 1:, :] - y_pred[..., :-1, :]
    dxg = y[..., :, 1:] - y[..., :, :-1]
    dyg = y[..., 1:, :] - y[..., :-1, :]
    gd = (dxp[..., :H - 1, :W - 1] - dxg[..., :H - 1, :W - 1]).abs() + \
         (dyp[..., :H - 1, :W - 1] - dyg[..., :H - 1, :W - 1]).abs()
    if masked:
        mc = mask[..., :H - 1, :W - 1]
        gl = (gd * mc).sum() / (mc.sum() + 1e-8)
    else:
        gl = gd.mean()
    return l1 + 0.005 * gl


# ---------------------------------------------------------------------------
# One optimisation step (main.py:91-108, :275)
# ---------------------------------------------------------------------------
def clip_grad_norm(grads: Dict[str, Tensor], max_norm: float = 1.0) -> Tuple[Dict[str, Tensor], Tensor]:
    """Global L2-norm clip as called at main.py:106 (coef = max_norm/(norm+1e-6), clamped to 1)."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).float()
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return {k: g * coef for k, g in grads.items()}, total


def adamw_step(params: Params, grads: Dict[str, Tensor], m: Params, v: Params, step: int,
               lr: float = 1e-3, wd: float = 1e-4, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8):
    """torch.optim.AdamW(lr=1e-3, weight_decay=1e-4) as constructed at main.py:275 (decoupled decay)."""
    out_p, out_m, out_v = {}, {}, {}
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    for k, g in grads.items():
        w = params[k] * (1.0 - lr * wd)
        m_k = b1 * m[k] + (1.0 - b1) * g
        v_k = b2 * v[k] + (1.0 - b2) * g * g
        denom = v_k.sqrt() / math.sqrt(bc2) + eps
        out_p[k] = w - (lr / bc1) * m_k / denom
        out_m[k], out_v[k] = m_k, v_k
    return out_p, out_m, out_v


def is_trainable(name: str) -> bool:
    return not (name.endswith("running_mean") or name.endswith("running_var")
                or name.endswith("num_batches_tracked"))


def train_step(p: Params, x: Tensor, y: Tensor, mask: Optional[Tensor], use_mask: bool,
               m: Optional[Params] = None, v: Optional[Params] = None, step: int = 1):
    """zero_grad -> forward -> stack -> loss -> backward -> clip(1.0) -> AdamW, main.py:91-108.

    Returns ``(loss, new_params_and_buffers, grads_before_clip, m, v, y_pred)``.
    """
    leaves = {k: t.detach().clone().requires_grad_(True) for k, t in p.items() if is_trainable(k)}
    full = {**{k: t for k, t in p.items() if not is_trainable(k)}, **leaves}
    buffers: Params = {}
    outs, _ = model_forward(full, x, None, True, buffers)
    y_pred = torch.stack(outs, dim=1)                       # main.py:98
    loss = compute_loss(y_pred, y, mask, use_mask)
    names = list(leaves.keys())
    gl = torch.autograd.grad(loss, [leaves[k] for k in names])
    grads = dict(zip(names, gl))
    clipped, _ = clip_grad_norm(grads, 1.0)
    if m is None:
        m = {k: torch.zeros_like(g) for k, g in grads.items()}
        v = {k: torch.zeros_like(g) for k, g in grads.items()}
    new_p, m, v = adamw_step({k: leaves[k].detach() for k in names}, clipped, m, v, step)
    out = dict(p)
    out.update(new_p)
    out.update(buffers)
    return loss.detach(), out, grads, m, v, y_pred.detach()


# ---------------------------------------------------------------------------
# Dataset transform (train/unet.py:273-323) -- "next" row 8f-2
# ---------------------------------------------------------------------------
def dataset_transform(x_raw: Tensor, y_raw: Tensor, norm_const: float, min_vel: float, max_vel: float,
                      y_scale: float, trans_min: float, trans_max: float, clip: bool = True):
    """mask from RAW x (>1.1) before scaling, x/norm_const, clip -> asinh(y/scale) -> [-1,1]; train/unet.py:279-299."""
    mask = (x_raw[:, 0:1] > 1.1).float()
    x = x_raw / norm_const
    yr = torch.clamp(y_raw, min_vel, max_vel) if clip else y_raw
    yt = torch.asinh(yr / y_scale)
    ys = 2 * (yt - trans_min) / (trans_max - trans_min) - 1.0
    return x, ys.float(), mask


def denormalize(y_norm: Tensor, y_scale: float, trans_min: float, trans_max: float) -> Tensor:
    """Inverse of the asinh normalisation, train/unet.py:316-319."""
    yt = (y_norm + 1.0) / 2.0 * (trans_max - trans_min) + trans_min
    return torch.sinh(yt) * y_scale
