"""The training step's tail around the UNet as fused HIP kernels: noise-prediction MSE with its gradient,
global gradient-norm clipping and AdamW.

The reference's trainer does these with torch (train_latent_diffusion.py): ``F.mse_loss(pred.float(), noise.float())``
:1108, ``accelerator.clip_grad_norm_(params, 1.0)`` :1123, ``torch.optim.AdamW(lr, betas=(0.9, 0.999), weight_decay=1e-2,
eps=1e-8)`` :909-915 / ``optimizer.step()`` :1124.  ``FusedAdamW`` is a ``torch.optim.Optimizer`` with AdamW's constructor,
``param_groups`` (LR schedulers keep working) and state layout (``step``, ``exp_avg``, ``exp_avg_sq``: state dicts
interoperate), whose ``step()`` is two multi-tensor launches for all 654 tensors: the gradient norm (deterministic two-stage
reduction) and the update, which reads the clip factor from device memory and scales the gradients on the fly -- no host
sync, and the gradients are read once instead of three times.  ``mse_loss`` returns the float32 loss and writes the gradient
of the prediction in the same pass.  float32 parameters on one device (the trainer's master weights).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib
from .ops import _gpu, _stream, dt_code


class _TensorTables:
    """Device tables of a tensor list for the multi-tensor kernels (include/ldm_amd.h)."""

    def __init__(self, numels, device):
        chunk = _lib.load().ldm_opt_chunk_elements()
        ct, co, cl = [], [], []
        for i, n in enumerate(numels):
            for off in range(0, n, chunk):
                ct.append(i)
                co.append(off)
                cl.append(min(chunk, n - off))
        self.nchunks = len(ct)
        self.ntensors = len(numels)
        self.chunk_tensor = torch.tensor(ct, dtype=torch.int32).pin_memory().to(device, non_blocking=True)
        self.chunk_off = torch.tensor(co, dtype=torch.int64).pin_memory().to(device, non_blocking=True)
        self.chunk_len = torch.tensor(cl, dtype=torch.int32).pin_memory().to(device, non_blocking=True)
        self.partial = torch.empty(max(self.nchunks, 1), dtype=torch.float32, device=device)

    _cache = {}

    @staticmethod
    def pointers(tensors, device, cache_key=None) -> torch.Tensor:
        """Device table of the tensors' addresses.  Uploaded from pinned memory without blocking the host (a pageable
        copy would synchronise and stop the host from queueing the next step ahead of the GPU); parameter / optimizer
        state tables are cached until an address changes."""
        ptrs = tuple(t.data_ptr() for t in tensors)
        if cache_key is not None:
            hit = _TensorTables._cache.get((cache_key, device))
            if hit is not None and hit[0] == ptrs:
                return hit[1]
        table = torch.tensor(ptrs, dtype=torch.int64).pin_memory().to(device, non_blocking=True)
        if cache_key is not None:
            _TensorTables._cache[(cache_key, device)] = (ptrs, table)
        return table


def _dense(t: torch.Tensor) -> bool:
    """Non-overlapping and dense: the elements fill one gap-free run of storage in SOME dimension order (contiguous,
    channels_last -- what ``module.to(memory_format=torch.channels_last)``, train_latent_diffusion.py:876, makes of every
    3x3 filter --, a flat bucket slice, ...).  The multi-tensor kernels are elementwise over raw storage, so that is all
    they need, provided parameter, gradient and optimizer state of one tensor share their strides."""
    if t.is_contiguous():
        return True
    expect = 1
    for stride, size in sorted((st, sz) for sz, st in zip(t.shape, t.stride()) if sz != 1):
        if stride != expect:
            return False
        expect *= size
    return True


def _check_f32(tensors, what):
    for t in tensors:
        if t.dtype != torch.float32 or not _dense(t):
            raise TypeError(f"ldm_amd.optim: {what} must be dense (non-overlapping) float32 tensors (the trainer's master weights)")
    _gpu(*tensors)


def _like_param(p: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """``t`` with the strides of ``p`` (a copy only when they differ: a contiguous gradient of a channels_last filter)."""
    if t.stride() == p.stride() or t.numel() <= 1:
        return t
    return torch.empty_like(p, memory_format=torch.preserve_format).copy_(t)


def clip_grad_norm_(parameters, max_norm: float) -> torch.Tensor:
    """torch.nn.utils.clip_grad_norm_ (L2): scales the .grad tensors in place so their global norm is at most max_norm and
    returns the norm before clipping (0-d device tensor).  Two fused launches for the reduction; prefer
    ``FusedAdamW(max_grad_norm=...)``, which folds the scaling into the update instead of rewriting the gradients."""
    params = [p for p in parameters if p.grad is not None]
    if not params:
        return torch.zeros(())
    grads = [p.grad for p in params]
    _check_f32(grads, "gradients")
    dev = grads[0].device
    tabs = _TensorTables([g.numel() for g in grads], dev)
    out = torch.empty(2, dtype=torch.float32, device=dev)
    gp = _TensorTables.pointers(grads, dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().ldm_grad_norm_clip(gp.data_ptr(), tabs.ntensors, tabs.chunk_tensor.data_ptr(), tabs.chunk_off.data_ptr(),
                                                  tabs.chunk_len.data_ptr(), tabs.nchunks, float(max_norm), tabs.partial.data_ptr(),
                                                  out.data_ptr(), _stream(grads[0])), "ldm_grad_norm_clip")
    torch._foreach_mul_(grads, out[1])
    return out[0]


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2,
                 max_grad_norm: Optional[float] = None):
        if lr < 0 or eps < 0 or weight_decay < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1:
            raise ValueError("FusedAdamW: invalid hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.max_grad_norm = max_grad_norm
        self.last_grad_norm: Optional[torch.Tensor] = None     # 0-d device tensor after step() when clipping
        self._tables = {}

    def _tables_for(self, key, numels, device):
        t = self._tables.get(key)
        if t is None or t[0] != numels:
            t = self._tables[key] = (numels, _TensorTables(numels, device))
        return t[1]

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        groups = []
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            _check_f32(ps, "parameters")
            for p in ps:
                if p.grad.dtype != torch.float32:
                    raise TypeError("ldm_amd.optim: gradients must be float32 (the trainer's master weights)")
                st = self.state[p]
                if not st:
                    st["step"] = torch.zeros((), dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                for k in ("exp_avg", "exp_avg_sq"):      # a loaded state dict may carry another layout
                    st[k] = _like_param(p, st[k])
            groups.append((gi, group, ps))
        if not groups:
            return loss
        dev = groups[0][2][0].device
        clip_ptr = 0
        gp_all = None
        with torch.cuda.device(dev):
            stream = _stream(groups[0][2][0])
            # gradients as the update kernel reads them: the parameter's own strides (elementwise over raw storage)
            gview = {p: _like_param(p, p.grad) for _, _, ps in groups for p in ps}
            _gpu(*gview.values())
            if self.max_grad_norm is not None:      # one norm over every group, as clip_grad_norm_(model.parameters()) does
                allg = [gview[p] for _, _, ps in groups for p in ps]
                tabs = self._tables_for("norm", tuple(g.numel() for g in allg), dev)
                gp = _TensorTables.pointers(allg, dev)
                gp_all = gp if len(groups) == 1 else None
                clip = torch.empty(2, dtype=torch.float32, device=dev)
                _lib.check(lib.ldm_grad_norm_clip(gp.data_ptr(), tabs.ntensors, tabs.chunk_tensor.data_ptr(), tabs.chunk_off.data_ptr(),
                                                  tabs.chunk_len.data_ptr(), tabs.nchunks, float(self.max_grad_norm),
                                                  tabs.partial.data_ptr(), clip.data_ptr(), stream), "ldm_grad_norm_clip")
                self.last_grad_norm = clip[0]
                clip_ptr = clip.data_ptr()
            for gi, group, ps in groups:
                steps = set()
                for p in ps:
                    self.state[p]["step"] += 1
                    steps.add(int(self.state[p]["step"]))
                if len(steps) != 1:
                    raise RuntimeError("FusedAdamW: parameters of one group must share their step count")
                tabs = self._tables_for(gi, tuple(p.numel() for p in ps), dev)
                pp = _TensorTables.pointers(ps, dev, (id(self), gi, "p"))
                gp = gp_all if gp_all is not None else _TensorTables.pointers([gview[p] for p in ps], dev)
                mp = _TensorTables.pointers([self.state[p]["exp_avg"] for p in ps], dev, (id(self), gi, "m"))
                vp = _TensorTables.pointers([self.state[p]["exp_avg_sq"] for p in ps], dev, (id(self), gi, "v"))
                b1, b2 = group["betas"]
                _lib.check(lib.ldm_adamw_step(pp.data_ptr(), gp.data_ptr(), mp.data_ptr(), vp.data_ptr(), tabs.ntensors,
                                              tabs.chunk_tensor.data_ptr(), tabs.chunk_off.data_ptr(), tabs.chunk_len.data_ptr(),
                                              tabs.nchunks, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                              float(group["weight_decay"]), steps.pop(), clip_ptr, stream), "ldm_adamw_step")
        # the kernels wrote the parameters behind torch's back: bump their version counters, which the UNet's packed-weight
        # cache (and autograd's saved-tensor checks) key on
        for _, _, ps in groups:
            for p in ps:
                torch.autograd.graph.increment_version(p)
        return loss


def mse_loss(pred: torch.Tensor, target: torch.Tensor, want_grad: bool = True):
    """(loss, dpred): loss = mean((pred.float() - target.float())^2) as a 0-d float32 tensor, dpred its gradient w.r.t.
    pred in pred's dtype (None if not wanted)."""
    _gpu(pred, target)
    if pred.shape != target.shape:
        raise ValueError("mse_loss: shapes differ")
    pred = pred.contiguous()
    target = target.to(torch.float32).contiguous()
    n = pred.numel()
    lib = _lib.load()
    loss = torch.empty((), dtype=torch.float32, device=pred.device)
    dpred = torch.empty_like(pred) if want_grad else None
    ws = torch.empty(int(lib.ldm_mse_workspace_floats(n)), dtype=torch.float32, device=pred.device)
    with torch.cuda.device(pred.device):
        _lib.check(lib.ldm_mse_loss(pred.data_ptr(), dt_code(pred.dtype), target.data_ptr(), n, loss.data_ptr(),
                                    0 if dpred is None else dpred.data_ptr(), ws.data_ptr(), _stream(pred)), "ldm_mse_loss")
    return loss, dpred
