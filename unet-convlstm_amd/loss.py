"""compute_loss of the reference's main.py:28-72 as one fused HIP reduction (+ one backward kernel)."""
from __future__ import annotations

import torch

from . import ops


def compute_loss(y_pred, y, mask=None, use_mask=True):
    """Weighted L1 (weight 1+4|y|^3) + 0.005 x spatial-gradient L1; same signature and semantics as
    main.py:28-72.  Inputs are ``[B,T,1,H,W]`` (any leading dims; gradients along the last two)."""
    return ops.LossFn.apply(y_pred.contiguous().float(), y, mask, bool(use_mask))
