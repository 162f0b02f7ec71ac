#!/usr/bin/env python3
"""Single-rank cost of the data-parallel wrapper (developer tool, GPU only).

    python tools/ddp_overhead.py plain|pg|pg_noid|ddp|ddp_nohooks

plain: no process group.  pg: RCCL communicator created eagerly (device_id=...), no wrapper.  pg_noid: lazy communicator.
ddp: FlatDDP over RCCL with one rank.  Prints ms per step, host enqueue time per step and how many candidate streams
ops.side_stream() had to try (an RCCL communicator shifts the runtime's stream -> hardware-queue assignment).
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
mode = sys.argv[1]
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
if mode in ("pg", "pg_noid", "ddp", "ddp_nohooks"):
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    if mode == "pg_noid":
        dist.init_process_group("nccl", rank=0, world_size=1)
    else:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import unet_convlstm_amd as U
torch.manual_seed(1234)
model = U.TemporalUNetDualView(1, 1, base_ch=64, lstm_layers=1, use_skip_lstm=True, use_attention=False).to(dev).train()
opt = U.FusedAdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, max_grad_norm=1.0)
ddp = None
if mode.startswith("ddp"):
    ddp = U.FlatDDP(model, opt.flat)
    if mode == "ddp_nohooks":
        ddp.remove_hooks()
data = U.SyntheticSequences(32, 20, 64, 64, seed=1, kind="uniform", device=dev)
x, y = data.x, data.y
for _ in range(3):
    U.train_step(model, opt, x, y, None, False, ddp)
torch.cuda.synchronize()
t0 = time.perf_counter(); cpu = 0.0
for _ in range(5):
    c0 = time.perf_counter()
    U.train_step(model, opt, x, y, None, False, ddp)
    cpu += time.perf_counter() - c0
torch.cuda.synchronize()
t1 = time.perf_counter()
from unet_convlstm_amd import ops
print("probes", ops._SIDE_STREAM_PROBES, end="  ")
print(f"{mode:12s} step {1e3*(t1-t0)/5:7.2f} ms   cpu enqueue {1e3*cpu/5:7.2f} ms", flush=True)
if dist.is_initialized():
    dist.destroy_process_group()
