"""unet-convlstm_amd: the UNet-ConvLSTM training path of dordanino12/unet-convlstm on MI355X (gfx950).

Hand-written HIP kernels (libuclstm.so, C ABI in include/uclstm.h) behind the reference's own
nn.Module surface.  Importing this package loads the shared library and fails loudly if it is
missing -- there is no CPU or eager-PyTorch fallback for the compute path.

The directory name contains a hyphen (it mirrors the reference repository's name); import it as
``import unet_convlstm_amd`` (top-level alias module) or ``importlib.import_module("unet-convlstm_amd")``.
"""
from . import _lib
from ._lib import UclstmError
from . import ops
from .modules import (ConvLSTMCell, ConvLSTM, DoubleConv, Down, Up, OutConv, SpatialAttention,
                      TemporalUNetDualView, UNet)
from .loss import compute_loss
from .optim import FusedAdamW
from .engine import train_one_epoch, evaluate, train_step, GraphedTrainStep, quiesce_host_gc, SyntheticSequences, NPZSequenceDataset, device_transform
from .ddp import FlatDDP
from .streaming import StreamingPredictor
from .ops import compute_dtype, set_compute_dtype, get_compute_dtype

__all__ = ["ConvLSTMCell", "ConvLSTM", "DoubleConv", "Down", "Up", "OutConv", "SpatialAttention",
           "TemporalUNetDualView", "UNet", "compute_loss", "FusedAdamW", "train_one_epoch", "evaluate",
           "train_step", "GraphedTrainStep", "quiesce_host_gc", "SyntheticSequences", "NPZSequenceDataset", "device_transform", "FlatDDP", "StreamingPredictor", "UclstmError", "ops",
           "compute_dtype", "set_compute_dtype", "get_compute_dtype"]
