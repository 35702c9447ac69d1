"""Data-parallel training of the UNet over RCCL / xGMI: bucketed gradient all-reduce overlapped with the backward.

The reference never names a collective: ``accelerator.prepare(model)`` (train_latent_diffusion.py:947) wraps the UNet in
torch DDP when launched with more than one process, and DDP's reducer all-reduces 25 MB gradient buckets as autograd
produces them (``accelerator.backward`` :1119, ``accumulate`` / ``no_sync`` :1032).  The HIP UNet is ONE autograd node
(ldm/_train.py), so torch DDP would see every gradient at the same instant, after the whole backward.  The overlap
therefore lives inside the hand-scheduled backward: ``Tape.put`` hands each finished parameter gradient to a
``GradReducer``, whose flat buckets ARE the gradients (as DDP's ``gradient_as_bucket_view``): the weight-gradient kernels
write their result straight into the parameter's bucket slice, ``.grad`` is set to that slice, and
``torch.distributed.all_reduce(bucket, AVG, async_op=True)`` -- RCCL on its own stream -- is launched as soon as a bucket
is full, while the backward kernels of earlier layers keep running.  The node's backward ends by waiting for the
outstanding buckets.  No gradient is copied on its way from the kernel to the optimizer.

xGMI is point-to-point (7 links per GPU): a ring all-reduce is bound by one link, so buckets are large (64 MiB default:
few, long transfers; 3.44 GB of float32 gradients = 52 buckets) and there is no attempt to mimic NCCL/NVSwitch tuning.

``DataParallel`` is the module-level wrapper (``.module``, ``no_sync()``, same ``forward``); with gradient accumulation
the unsynchronised micro-batches accumulate in ``.grad`` locally and the synchronising backward reduces the sum.
"""
from __future__ import annotations

import contextlib
from typing import Dict, List, Optional

import torch
import torch.distributed as dist
from torch import nn


class _Done:
    def wait(self):
        return True


def _host_staged(group) -> bool:
    """gloo cannot move device memory on this platform (its CUDA-tensor collectives never complete on ROCm): device tensors
    take a synchronous round trip through host memory.  Rehearsal path only -- several ranks sharing one GPU, CPU-only
    launches; RCCL ("nccl"), the backend of record, reduces device memory in place, asynchronously."""
    return dist.get_backend(group) != "nccl"


def _all_reduce(t: torch.Tensor, op, group):
    if t.is_cuda and _host_staged(group):
        host = t.cpu()
        dist.all_reduce(host, op=op, group=group)
        t.copy_(host)
        return _Done()
    return dist.all_reduce(t, op=op, group=group, async_op=True)


def _broadcast(t: torch.Tensor, src: int, group):
    if t.is_cuda and _host_staged(group):
        host = t.cpu()
        dist.broadcast(host, src=src, group=group)
        t.copy_(host)
    else:
        dist.broadcast(t, src=src, group=group)


class GradReducer:
    """Flat-bucket gradient averaging across a process group, fed in gradient-arrival order.

    Bucket layout is fixed by the order in which gradients arrive during the first synchronised backward (identical on
    every rank: the schedule is deterministic), like DDP's bucket rebuild; that first pass reduces bucket by bucket at
    the end, later passes overlap."""

    def __init__(self, params: Dict[str, torch.Tensor], process_group=None, bucket_bytes: int = 64 << 20,
                 always_reduce: bool = False):
        self.params = params                      # name -> nn.Parameter
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # always_reduce: launch the collectives also in a group of ONE rank (an identity).  For tests and rehearsals: a
        # one-GPU box then drives the real RCCL path -- async all-reduce of bucket views on RCCL's stream while the
        # backward kernels keep running, handle waits at the end -- which world == 1 otherwise skips.
        self.always_reduce = bool(always_reduce) and dist.is_initialized()
        self.bucket_bytes = int(bucket_bytes)
        self.order: List[str] = []                # arrival order (first pass)
        self.slot: Dict[str, tuple] = {}          # name -> (bucket index, offset, numel)
        self.buckets: List[torch.Tensor] = []
        self.fill: List[int] = []                 # elements received per bucket in the current pass
        self.size: List[int] = []
        self.handles: List = []
        self.pending: Dict[str, torch.Tensor] = {}
        self.launched = 0                         # collectives launched in the current pass (tests / bench)
        self._post_scale: List[int] = []

    # -------------------------------------------------------------------------------------------- layout
    def _build(self, device, dtype):
        cur, cur_elems = [], 0
        groups = []
        for name in self.order:
            n = self.params[name].numel()
            if cur and (cur_elems + n) * 4 > self.bucket_bytes:
                groups.append(cur)
                cur, cur_elems = [], 0
            cur.append(name)
            cur_elems += n
        if cur:
            groups.append(cur)
        for bi, names in enumerate(groups):
            off = 0
            for name in names:
                n = self.params[name].numel()
                self.slot[name] = (bi, off, n)
                off += (n + 63) // 64 * 64        # 256-byte aligned slices
            self.buckets.append(torch.zeros(off, dtype=dtype, device=device))
            self.size.append(len(names))
        self.fill = [0] * len(self.buckets)

    # ---------------------------------------------------------------------------------------------- pass
    def begin(self):
        self.handles, self.pending, self.launched = [], {}, 0
        self.fill = [0] * len(self.buckets)
        self._post_scale: List[int] = []

    def _view(self, name: str) -> torch.Tensor:
        bi, off, n = self.slot[name]
        return self.buckets[bi][off:off + n]

    def dest(self, name: str, shape) -> Optional[torch.Tensor]:
        """Zero-copy path (DDP's ``gradient_as_bucket_view``): the bucket slice of ``name``, shaped ``shape``, for a
        producer that can write this pass's gradient in place and then hands the same tensor to ``put``.  None when the
        gradient has to be combined with an existing ``.grad`` or the layout is not fixed yet."""
        if name not in self.slot or self.params[name].grad is not None:
            return None
        return self._view(name).view(shape)

    def owns(self, name: str, g: torch.Tensor) -> bool:
        return name in self.slot and g.dtype == torch.float32 and g.data_ptr() == self._view(name).data_ptr()

    def put(self, name: str, g: torch.Tensor) -> torch.Tensor:
        """Takes this rank's finished gradient of parameter ``name`` (any shape with the parameter's element order;
        strided views are copied once, straight into the bucket); returns the tensor that will hold the average once
        ``finish()`` has run."""
        p = self.params[name]
        acc = p.grad                               # micro-batches accumulated under no_sync()
        if acc is not None:
            p.grad = None                          # the reduced sum becomes the gradient of this backward
        if not self.slot:                          # first pass: record the order, reduce at the end
            self.order.append(name)
            g = g.reshape(p.shape).to(torch.float32)
            self.pending[name] = g if acc is None else g + acc.to(torch.float32)
            return self.pending[name]
        if name not in self.slot:                  # a parameter that had no gradient in the first pass
            g = g.reshape(p.shape).to(torch.float32)
            self.pending[name] = (g if acc is None else g + acc.to(torch.float32)) * (1.0 / self.world)
            return self.pending[name]
        bi = self.slot[name][0]
        flat = self._view(name)
        if acc is None:
            if not (g.dtype == torch.float32 and g.data_ptr() == flat.data_ptr()):   # else: written in place (dest)
                flat.view(g.shape).copy_(g)
        else:                                      # acc may be this very slice (grads zeroed in place, or never zeroed)
            torch.add(g.reshape(p.shape), acc.reshape(p.shape), out=flat.view(p.shape))
        self.fill[bi] += 1
        if self.fill[bi] == self.size[bi]:
            self._launch(bi)
        return flat.view(p.shape)

    def _launch(self, bi: int, prescaled: bool = False):
        """Average bucket ``bi`` over the group.  RCCL averages in the collective; backends without AVG (gloo) sum and
        the bucket is scaled when the pass finishes."""
        self.launched += 1
        if self.world > 1 or self.always_reduce:
            op = dist.ReduceOp.SUM
            if not prescaled:
                if dist.get_backend(self.group) == "nccl":
                    op = dist.ReduceOp.AVG
                else:
                    self._post_scale.append(bi)
            self.handles.append(_all_reduce(self.buckets[bi], op, self.group))

    def finish(self) -> Dict[str, torch.Tensor]:
        """Completes the pass: after this the tensors returned by put() hold the averaged gradients (stream-ordered on
        the current stream).  Returns replacements for first-pass gradients (which were not bucket views yet)."""
        replaced: Dict[str, torch.Tensor] = {}
        if not self.slot and self.order:
            any_g = next(iter(self.pending.values()))
            self._build(any_g.device, torch.float32)
            for name in self.order:
                view = self._view(name).view(self.pending[name].shape)
                torch.mul(self.pending[name], 1.0 / self.world, out=view)
                replaced[name] = view
            self.pending = {}
            for bi in range(len(self.buckets)):
                self._launch(bi, prescaled=True)
        else:
            for bi in range(len(self.buckets)):   # parameters without a gradient this pass leave a bucket short
                if 0 < self.fill[bi] < self.size[bi]:
                    self._launch(bi)
            for name in sorted(self.pending):      # strays (see put): one small collective each, same order everywhere
                self.launched += 1
                if self.world > 1 or self.always_reduce:
                    self.handles.append(_all_reduce(self.pending[name], dist.ReduceOp.SUM, self.group))
            self.pending = {}
        for h in self.handles:
            h.wait()
        self.handles = []
        for bi in self._post_scale:
            self.buckets[bi].mul_(1.0 / self.world)
        self._post_scale = []
        return replaced


class DataParallel(nn.Module):
    """Drop-in for the DDP wrap accelerate applies to the UNet: ``DataParallel(unet)(sample, t, ctx)``; gradients in
    ``unet.parameters()`` after ``loss.backward()`` are the average over the process group."""

    def __init__(self, module: nn.Module, process_group=None, bucket_mb: float = 64.0, always_reduce: bool = False):
        super().__init__()
        self.module = module
        self.reducer = GradReducer(dict(module.named_parameters()), process_group, int(bucket_mb * (1 << 20)),
                                   always_reduce=always_reduce)
        self.require_sync = True
        if dist.is_initialized() and dist.get_world_size(process_group) > 1:
            with torch.no_grad():                  # replicas start from rank 0's weights, as DDP's constructor does
                for p in module.parameters():
                    _broadcast(p.data, dist.get_global_rank(process_group, 0) if process_group else 0, process_group)

    @contextlib.contextmanager
    def no_sync(self):
        """Gradient accumulation (train_latent_diffusion.py:1032): backward passes inside do not communicate."""
        old, self.require_sync = self.require_sync, False
        try:
            yield
        finally:
            self.require_sync = old

    def forward(self, *args, **kwargs):
        self.module._ldm_grad_sink = self.reducer if self.require_sync else None
        try:
            return self.module(*args, **kwargs)
        finally:
            self.module._ldm_grad_sink = None
