"""Data-parallel training across the GPUs of one node: one process per GPU, sequences sharded over
ranks, one gradient exchange per step.

``FlatDDP`` all-reduces the flat gradient buffer of ``optim.FlatParams`` in contiguous buckets cut
from the END of the buffer backwards: backward produces gradients in (nearly) reverse registration
order -- decoder first, then the three ConvLSTM weights when their BPTT finishes, encoder last -- so
each bucket is launched (async, on RCCL's own stream over xGMI) as soon as its last gradient has
been accumulated and overlaps the rest of backward.  BatchNorm statistics stay local per rank
(the reference has no SyncBN); parameters and buffers are broadcast from rank 0 once.

Backend-agnostic: ``nccl`` (= RCCL on ROCm) on GPUs, ``gloo`` in the CPU tests.
"""
from __future__ import annotations

import weakref
from typing import List, Optional

import torch
import torch.distributed as dist

from .optim import FlatParams


def _unregister(side_hook, use_hook) -> None:
    from . import ops
    if side_hook in ops.GRAD_SIDE_HOOKS:
        ops.GRAD_SIDE_HOOKS.remove(side_hook)
    if use_hook in ops.USE_HOOKS:
        ops.USE_HOOKS.remove(use_hook)


class FlatDDP:
    def __init__(self, module: torch.nn.Module, flat: FlatParams, bucket_mb: float = 64.0, process_group=None,
                 broadcast: bool = True, first_bucket_mb: float = 1.0, grad_dtype: Optional[torch.dtype] = None):
        """``grad_dtype=torch.bfloat16``: exchange the gradients as bf16 (half the bytes per link; every bucket is cast into
        a staging buffer, all-reduced there and cast back into the f32 gradient buffer before the optimiser step)."""
        if not dist.is_initialized():
            raise RuntimeError("FlatDDP: torch.distributed is not initialised")
        if grad_dtype not in (None, torch.float32, torch.bfloat16):
            raise ValueError("FlatDDP: grad_dtype must be None / torch.float32 / torch.bfloat16")
        owner = getattr(flat, "_ddp_owner", None)
        if owner is not None and owner() is not None and owner()._hooks:
            # two wrappers on one gradient buffer would each all-reduce every bucket: the gradients would be averaged twice
            raise RuntimeError("FlatDDP: these FlatParams already belong to a live FlatDDP; call remove_hooks() on it first")
        flat._ddp_owner = weakref.ref(self)
        self.module, self.flat, self.pg = module, flat, process_group
        self.grad_dtype = None if grad_dtype in (None, torch.float32) else grad_dtype
        self._stage = torch.empty_like(flat.flat_g, dtype=self.grad_dtype) if self.grad_dtype is not None else None
        self.world = dist.get_world_size(process_group)
        # RCCL averages inside the collective; gloo (CPU tests) has no AVG and gets an explicit scale in finalize()
        self._op = dist.ReduceOp.AVG if dist.get_backend(process_group) == "nccl" else dist.ReduceOp.SUM
        if broadcast:
            dist.broadcast(flat.flat_p, src=0, group=process_group)
            for b in module.buffers():
                dist.broadcast(b, src=0, group=process_group)
        # buckets: contiguous [start, end) slices, last parameters first.  The bucket that holds parameter 0 completes only
        # with the very last gradient of the backward pass, so its all-reduce is the one nothing can hide: it is kept small
        # (first_bucket_mb), like the first bucket of torch's DistributedDataParallel.
        cap = max(1, int(bucket_mb * (1 << 20) / 4))
        first_cap = min(cap, max(1, int(first_bucket_mb * (1 << 20) / 4)))
        n_front, size = 0, 0
        while n_front < len(flat.params) - 1 and size + flat.params[n_front].numel() <= first_cap:
            size += flat.params[n_front].numel()
            n_front += 1
        n_front = max(n_front, 1) if len(flat.params) > 1 else len(flat.params)
        self.buckets: List[List[int]] = []          # parameter indices per bucket
        self.ranges: List[tuple] = []

        def close(cur):
            lo = flat.offsets[cur[-1]]
            hi = flat.offsets[cur[0]] + flat.params[cur[0]].numel()
            self.buckets.append(cur)
            self.ranges.append((lo, hi))

        cur: List[int] = []
        size = 0
        for i in range(len(flat.params) - 1, n_front - 1, -1):
            cur.append(i)
            size += flat.params[i].numel()
            if size >= cap or i == n_front:
                close(cur)
                cur, size = [], 0
        if n_front > 0 and len(flat.params) > 0:
            close(list(range(n_front - 1, -1, -1)))
        self.bucket_of = {}
        for bi, idxs in enumerate(self.buckets):
            for i in idxs:
                self.bucket_of[i] = bi
        self._pending = [0] * len(self.buckets)
        self._handles: List[Optional[object]] = [None] * len(self.buckets)
        self._hooks = []
        for i, p in enumerate(flat.params):
            self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))
        # weight gradients that the operators accumulate on their side stream never pass through autograd's accumulator:
        # they announce themselves through ops.GRAD_SIDE_HOOKS (called with the side stream current)
        self._index_of = {id(p): i for i, p in enumerate(flat.params)}
        self._uses = [0] * len(flat.params)          # uses reported by the operators' forward passes since reset()
        self._seen = [0] * len(flat.params)          # side announcements received since reset()
        self._done = [False] * len(flat.params)
        self._sync = True
        from . import ops
        # The operator layer keeps its hook lists in module globals: register through a weak reference, so that a wrapper
        # that is dropped without remove_hooks() takes its hooks with it instead of all-reducing behind its successor.
        me = weakref.ref(self)

        def side_hook(param):
            o = me()
            if o is None:
                _unregister(side_hook, use_hook)
            elif id(param) in o._index_of:
                o._on_ready(o._index_of[id(param)], True)

        def use_hook(param):
            o = me()
            if o is None:
                _unregister(side_hook, use_hook)
            else:
                o._on_use(param)

        self._side_hook, self._use_hook = side_hook, use_hook
        ops.GRAD_SIDE_HOOKS.append(side_hook)
        ops.USE_HOOKS.append(use_hook)
        weakref.finalize(self, _unregister, side_hook, use_hook)
        self._main_stream = None
        self.reset()

    def describe(self) -> dict:
        """What one step exchanges: bucket count, all-reduce bytes per step and the dtype on the wire."""
        el = 2 if self.grad_dtype is not None else 4
        return {"buckets": len(self.buckets), "allreduce_bytes": int(self.flat.numel) * el,
                "dtype": "bf16" if self.grad_dtype is not None else "f32", "world_size": self.world,
                "bucket_bytes": [int(hi - lo) * el for lo, hi in self.ranges]}

    def _on_use(self, param) -> None:
        i = self._index_of.get(id(param))
        if i is None:
            return
        if self._sync:
            self._uses[i] += 1
        else:
            self._unsynced_uses[i] += 1

    def _on_ready(self, i: int, side: bool = False) -> None:
        """Gradient ``i`` has been produced.  ``side``: announced by an operator that wrote ``.grad`` itself -- once per USE
        of the parameter, so the parameter is complete only after as many announcements as forward reported uses;
        autograd's own accumulator (``side=False``) fires once per backward pass, after all uses."""
        if not self._sync:
            if side and self._unsynced_uses[i] > 0:
                self._unsynced_uses[i] -= 1          # a pass that ran entirely inside no_sync() pays its own uses off
            return
        if side:
            if self._unsynced_uses[i]:
                # some of this parameter's uses were recorded inside no_sync() (or a forward ran there and its backward runs
                # here): the use count of THIS pass is unknown, so nothing is released early -- finalize() launches the bucket
                return
            if self._uses[i] == 0:
                # a gradient for a parameter whose forward was never reported (forward before reset(), or by code that does
                # not call ops.note_use): completeness cannot be told from announcements; leave it to finalize()
                return
            self._seen[i] += 1
            if self._seen[i] > self._uses[i]:
                raise RuntimeError(f"FlatDDP: parameter {i} announced {self._seen[i]} gradients for {self._uses[i]} recorded uses "
                                   "(call reset() before every forward pass; do not mix passes of different steps)")
            if self._seen[i] < self._uses[i]:
                return
        if self._done[i]:
            return
        self._done[i] = True
        bi = self.bucket_of[i]
        self._pending[bi] -= 1
        if self._pending[bi] < 0:
            raise RuntimeError(f"FlatDDP: bucket {bi} completed more gradients than it holds (reset() missing before this step?)")
        if self._pending[bi] == 0:
            self._launch(bi)

    def _make_hook(self, i: int):
        def hook(_param):
            self._on_ready(i)
        return hook

    class _NoSync:
        def __init__(self, owner):
            self.owner = owner

        def __enter__(self):
            self.prev, self.owner._sync = self.owner._sync, False

        def __exit__(self, *exc):
            self.owner._sync = self.prev

    def no_sync(self):
        """Gradient accumulation: forward/backward passes inside ``with ddp.no_sync():`` neither count uses nor launch
        collectives; the gradients of the last pass (outside the context) are exchanged with everything accumulated."""
        return FlatDDP._NoSync(self)

    def _launch(self, bi: int) -> None:
        lo, hi = self.ranges[bi]
        g = self.flat.flat_g
        if not g.is_cuda:
            buf = g[lo:hi]
            if self._stage is not None:
                buf = self._stage[lo:hi]
                buf.copy_(g[lo:hi])
            self._handles[bi] = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            return
        # a bucket mixes gradients produced on the main stream (autograd accumulation) and on the operators' side stream
        # (weight-gradient GEMMs).  The collective is enqueued from a third, kernel-less stream that waits for both, so
        # neither of the two compute streams stalls on the other; RCCL's own stream orders itself after that stream.
        from . import ops
        main = self._main_stream or torch.cuda.current_stream(g.device)
        launch = ops.launch_stream(g.device, main)
        launch.wait_stream(main)
        launch.wait_stream(ops.side_stream(g.device))
        cur = torch.cuda.current_stream(g.device)
        if cur != main:
            launch.wait_stream(cur)
        with torch.cuda.stream(launch):
            buf = g[lo:hi]
            if self._stage is not None:
                buf = self._stage[lo:hi]
                buf.copy_(g[lo:hi])                     # f32 -> bf16 on the launch stream, right before the collective
            self._handles[bi] = dist.all_reduce(buf, op=self._op, group=self.pg, async_op=True)

    def reset(self) -> None:
        """Call once per optimisation step BEFORE the forward pass (after zero_grad), on the stream that will run the step:
        the operators report parameter uses during forward."""
        self._pending = [len(b) for b in self.buckets]
        self._handles = [None] * len(self.buckets)
        self._uses = [0] * len(self.flat.params)
        self._unsynced_uses = [0] * len(self.flat.params)
        self._seen = [0] * len(self.flat.params)
        self._done = [False] * len(self.flat.params)
        if self.flat.flat_g.is_cuda:
            self._main_stream = torch.cuda.current_stream(self.flat.flat_g.device)

    def finalize(self) -> None:
        """Call after backward, before the optimiser step: launches buckets whose hooks did not all fire
        (parameters unused in this step), waits for every collective and turns sums into means."""
        for bi in range(len(self.buckets)):
            if self._handles[bi] is None:
                self._launch(bi)
        for h in self._handles:
            h.wait()                   # the current stream waits for the collective's stream (no host block on GPUs)
        if self._stage is not None:
            self.flat.flat_g.copy_(self._stage)          # bf16 -> f32, one pass over the gradient buffer
        if self.world > 1 and self._op == dist.ReduceOp.SUM:
            self.flat.flat_g.mul_(1.0 / self.world)

    def remove_hooks(self) -> None:
        for h in self._hooks:
            h.remove()
        self._hooks = []
        if getattr(self, "_side_hook", None) is not None:
            _unregister(self._side_hook, self._use_hook)
            self._side_hook = self._use_hook = None

    def __call__(self, *a, **k):
        return self.module(*a, **k)
