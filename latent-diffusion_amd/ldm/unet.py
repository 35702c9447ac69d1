"""SD-1.5-shaped conditional UNet denoiser whose forward runs on MI355X HIP kernels.

Drop-in for ``ldm/unet.py`` of anenbergb/latent-diffusion: same public classes, constructor arguments and defaults
(unet.py:10-22), same ``forward(sample, timestep, encoder_hidden_states)`` (:118-156), same ``state_dict()`` key names
and shapes (so the reference's checkpoints load), default PyTorch parameter initialisation.  The torch layers inside
the modules below exist only to own the parameters under the reference's names; none of them is ever called.  The
arithmetic is a straight-line schedule of C-ABI kernel launches (``ops.py`` -> ``libldm_amd.so``):

* activations are NHWC in the compute dtype (bf16 with fp32 accumulation, or exact fp32), so the
  NCHW<->(B, HW, C) rearranges of TransformerModel (:509-520) vanish and 1x1 convs are plain GEMMs;
* every conv / linear is the MFMA implicit-GEMM kernel with bias, time-embedding add, residual add, SiLU and
  GEGLU/SwiGLU fused as epilogues; ``torch.cat`` of the up path (:313, :377) and ``F.interpolate`` (:316, :385)
  are folded into the operand gather; the 22 per-block time projections (:703, :727) run as ONE GEMM;
* q and kv projections of self-attention are one GEMM; attention is flash-style on the matrix cores.

The reference's quirks are reproduced: no skip around TransformerModel, no residual around
the feed-forward, attn2 always 8 heads, K = first half of kv_proj, GEGLU value = first half, GN/LN eps 1e-5.

There is no CPU fallback.  With autograd recording (the trainer's loss.backward(), train_latent_diffusion.py:1119) the
call runs the training form of the same schedule (``_train.py``): one autograd node per module call whose backward is
the hand-scheduled chain of HIP backward kernels; parameter gradients arrive as ordinary ``.grad`` tensors.
"""
from __future__ import annotations

import os
from typing import List, Optional, Union

import torch
from torch import nn
from torch.optim.optimizer import register_optimizer_step_post_hook

from . import _train, ops


def _j(prefix: str, name: str) -> str:
    return f"{prefix}.{name}" if prefix else name


def _seq(*mods):
    return nn.Sequential(*mods)


_LN_FOLD = os.environ.get("LDM_LN_FOLD", "1") != "0"      # A/B switch (tools): 0 = the separate LayerNorm kernel everywhere


def compute_dtype_of(x: torch.Tensor) -> torch.dtype:
    """bf16 storage / fp32 accumulation for bf16 inputs or under bf16 autocast, exact fp32 otherwise."""
    if x.dtype == torch.bfloat16:
        return torch.bfloat16
    if torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16:
        return torch.bfloat16
    if x.dtype == torch.float32:
        return torch.float32
    raise TypeError(f"ldm_amd: unsupported dtype {x.dtype} (float32 or bfloat16)")


# --------------------------------------------------------------------------------------- packed weights
# Optimizers that update parameters through fused multi-tensor kernels (torch.optim.AdamW(fused=True)) do not bump the
# parameters' version counters, so the packed-weight cache also keys on a global count of optimizer steps.
_optimizer_steps = [0]


def _count_optimizer_step(*_args, **_kwargs):
    _optimizer_steps[0] += 1


register_optimizer_step_post_hook(_count_optimizer_step)


def param_state_key(module: nn.Module):
    """Changes whenever a parameter of ``module`` may have changed: in-place ops and copies (version counters), moved
    storage (addresses), any optimizer step.  Writes through ``.data`` / raw pointers that bypass all three need
    ``invalidate_packed_weights()``."""
    return (_optimizer_steps[0], tuple((p._version, p.data_ptr()) for p in module.parameters()))


class Repack:
    """What ``pack_tree(train=True)`` leaves under ``P["_repack"]``: everything needed to rewrite the derived tensors IN PLACE
    after an optimizer step -- the batched filter pack (``ops.PackPlan``, one launch per tap count instead of one per filter)
    and the few derived copies that are not filters (packed GLU biases, the fused time-projection bias, casts of parameters
    that are not float32)."""

    def __init__(self, plan, refresh):
        self.plan, self.refresh = plan, refresh

    def run(self):
        self.plan.run()
        with torch.no_grad(), torch.autocast("cuda", enabled=False):
            for f in self.refresh:
                f()

    def still_valid(self) -> bool:
        return self.plan.still_valid()


def pack_tree(root: nn.Module, dtype: torch.dtype, train: bool = False) -> dict:
    """Derived, K-contiguous copies of every filter under ``root`` in the compute dtype, keyed by module path.
    Checkpoint layout stays PyTorch's ([out, in, kh, kw]); these are rebuilt when a parameter changes.
    train: also the filters of the input-gradient pass (``.wd``: in/out transposed, taps rotated), produced by the same
    pass over the checkpoint tensor -- all filters of the tree in one batched launch per tap count (``ops.PackPlan``), which
    ``P["_repack"].run()`` repeats in place after every optimizer step."""
    P = {}
    jobs = []           # train: (checkpoint filter, fwd destination, dgrad destination, glu_pairs)
    refresh = []        # train: closures that rewrite the derived NON-filter copies in place

    def f32(p):
        t = p.detach().to(torch.float32).contiguous()
        if t.data_ptr() != p.data_ptr():          # a real copy (parameter not float32 / not dense): must follow the parameter
            refresh.append(lambda t=t, p=p: t.copy_(p.detach()))
        return t

    def pair(w, glu=False, fwd=None, dgrad=None):
        """(fwd, dgrad-as-filter) of one checkpoint filter; packed now (inference) or by the batched plan (train)."""
        wd_ = w.detach()
        o, i = wd_.shape[0], wd_.shape[1]
        taps = 1 if wd_.dim() == 2 else wd_.shape[2] * wd_.shape[3]
        k = taps * i
        if fwd is None:
            fwd = torch.empty((o, k), dtype=dtype, device=w.device)
        if dgrad is None:
            dgrad = torch.empty((k, o), dtype=dtype, device=w.device)
        jobs.append((w, fwd, dgrad, glu))
        return fwd, (dgrad.view(i, taps * o) if dgrad.is_contiguous() else dgrad)

    def pack_fused(key, weights):
        """Several filters sharing their input, stacked along the output dimension, as one filter."""
        rows = sum(w.shape[0] for w in weights)
        k = weights[0][0].numel()
        dev = weights[0].device
        fwd = torch.empty((rows, k), dtype=dtype, device=dev)
        dg = torch.empty((k, rows), dtype=dtype, device=dev) if train else None
        r = 0
        for w in weights:
            n = w.shape[0]
            if train:
                pair(w, fwd=fwd[r:r + n], dgrad=dg[:, r:r + n])
            else:
                ops.pack_weight_pair(w, dtype, fwd=fwd[r:r + n], dgrad=None, want_dgrad=False)
            r += n
        P[key + ".w"] = fwd
        if train:
            P[key + ".wd"] = dg

    with torch.no_grad(), torch.autocast("cuda", enabled=False):   # packing may run inside the caller's autocast region
        for name, mod in root.named_modules():
            if getattr(mod, "_ldm_raw", False):       # conv_in / final conv keep float32 OIHW filters
                P[name + ".w"] = f32(mod.weight)
                P[name + ".b"] = f32(mod.bias)
            elif getattr(mod, "_ldm_skip", False) or (getattr(mod, "_ldm_skip_in_unet", False) and isinstance(root, UNet)):
                continue                              # covered by a fused filter below
            elif isinstance(mod, (nn.Conv2d, nn.Linear)):
                glu = name.endswith("w1_w2")
                if train:
                    P[name + ".w"], P[name + ".wd"] = pair(mod.weight, glu)
                else:
                    P[name + ".w"] = ops.pack_weight(mod.weight, dtype, glu_pairs=glu)
                if mod.bias is not None:
                    if glu:
                        gb = ops.pack_weight(mod.bias, torch.float32, glu_pairs=True)
                        refresh.append(lambda gb=gb, bias=mod.bias: ops.pack_weight(bias, torch.float32, glu_pairs=True, out=gb))
                        P[name + ".b"] = gb.flatten()
                    else:
                        P[name + ".b"] = f32(mod.bias)
            elif isinstance(mod, (nn.GroupNorm, nn.LayerNorm)):
                P[name + ".g"] = f32(mod.weight)
                P[name + ".b"] = f32(mod.bias)
            if isinstance(mod, Attention) and mod.fused_qkv:
                pack_fused(_j(name, "qkv"), [mod.q_proj.weight, mod.kv_proj.weight])
        if not train:
            # inference: the three LayerNorms of every TransformerBlock inside a TransformerModel (unet.py:550-551, :576)
            # folded into the GEMMs around them (ops.fold_layernorm; include/ldm_amd.h LDM_EPI_LNFOLD)
            for name, mod in root.named_modules():
                if isinstance(mod, TransformerModel) and _LN_FOLD and ops.ln_fold_supported(mod.in_channels, dtype) \
                        and ops.ln_fold_selfcheck(mod.norm.weight.device):
                    for k, blk in enumerate(mod.transformer_blocks):
                        b = _j(name, f"transformer_blocks.{k}")
                        a1 = blk.attn1
                        wqkv = torch.cat([a1.q_proj.weight, a1.kv_proj.weight], 0) if a1.fused_qkv else a1.q_proj.weight
                        for key, w, bias, ln, glu in (
                                (b + (".attn1.qkv" if a1.fused_qkv else ".attn1.q_proj"), wqkv, None, blk.norm1, False),
                                (b + ".attn2.q_proj", blk.attn2.q_proj.weight, None, blk.norm2, False),
                                (b + ".feed_forward.1.w1_w2", blk.feed_forward[1].w1_w2.weight, blk.feed_forward[1].w1_w2.bias,
                                 blk.feed_forward[0], True)):
                            P[key + ".w_ln"], P[key + ".s1"], P[key + ".b_ln"] = ops.fold_layernorm(w, bias, ln.weight, ln.bias,
                                                                                                  dtype, glu_pairs=glu)
                    P[_j(name, "ln_fold")] = True
        resnets = [(n, m) for n, m in root.named_modules() if isinstance(m, ResnetBlock)]
        if isinstance(root, UNet) and resnets:
            # all time projections (unet.py:703) as one filter; each resnet reads its column slice
            pack_fused("time_all", [m.time[1].weight for _, m in resnets])
            tb = torch.cat([m.time[1].bias.detach().to(torch.float32) for _, m in resnets], 0)
            refresh.append(lambda tb=tb, ms=[m for _, m in resnets]:
                           torch.cat([m.time[1].bias.detach().to(torch.float32) for m in ms], 0, out=tb))
            P["time_all.b"] = tb
            off = 0
            for n, m in resnets:
                P[n + ".time_off"] = off
                off += m.out_channels
        cross = [(n, m) for n, m in root.named_modules() if isinstance(m, Attention) and not m.fused_qkv]
        if isinstance(root, UNet) and cross and not train:
            # inference: the K|V projections of the text context of every cross-attention (unet.py:634, :656; 16 GEMMs
            # with 2,464 rows each, none of which depends on the image path) as one filter; each layer reads its
            # column slice of the (B, 77, sum 2C) result
            pack_fused("ctx_kv_all", [m.kv_proj.weight for _, m in cross])
            off = 0
            for n, m in cross:
                P[n + ".kv_off"] = off
                off += m.kv_proj.weight.shape[0]
        if train:
            plan = ops.PackPlan(jobs, dtype)
            plan.run()
            P["_repack"] = Repack(plan, refresh)
    return P


class _ContextKV:
    """The text context together with the K|V projections of all cross-attention layers (one GEMM, `ctx_kv_all`)."""

    def __init__(self, P, ctx):
        self.ctx = ctx
        self.kv_all = ops.conv_gemm(ctx, P["ctx_kv_all.w"])

    def of(self, P, name, width):
        off = P.get(name + ".kv_off")
        if off is None:      # a cross-attention as wide as the context keeps its own projection
            return ops.conv_gemm(self.ctx, P[_j(name, "kv_proj.w")])
        return self.kv_all[..., off:off + width]


class _HipModule(nn.Module):
    """Caches the packed weights of the tree rooted here until any parameter changes (optimizer step, load)."""

    def _packed_for(self, dtype: torch.dtype, train: bool = False) -> dict:
        ver = param_state_key(self)
        cache = self.__dict__.setdefault("_ldm_cache", {})
        if cache.get("ver") != ver:
            # a parameter changed.  The training set of the dtype being asked for is rewritten IN PLACE by its batched repack
            # (same buffers, two launches) when the parameters still live where its job tables point; everything else is dropped.
            kept = cache.get((dtype, True)) if train else None
            rp = kept.get("_repack") if kept is not None else None
            cache.clear()
            cache["ver"] = ver
            if rp is not None and rp.still_valid():
                rp.run()
                cache[(dtype, True)] = kept
        key = (dtype, train)
        if key not in cache:
            if train and (dtype, False) in cache:
                cache.pop((dtype, False))      # the training set contains the inference set
            cache[key] = pack_tree(self, dtype, train)
        return cache[key]

    def to(self, *args, **kwargs):
        """``nn.Module.to`` minus ``memory_format``: the reference trainer moves the UNet to channels_last
        (train_latent_diffusion.py:876) so that cuDNN runs NHWC; here every kernel already reads NHWC activations and derived
        K-contiguous filters (``pack_tree``), so the parameters keep the checkpoint's layout -- re-striding them would only
        add a copy per filter per step (pack, gradient, optimizer).  Shapes, values and ``state_dict()`` are unaffected."""
        kwargs.pop("memory_format", None)
        if not args and not kwargs:
            return self
        return super().to(*args, **kwargs)

    def invalidate_packed_weights(self):
        """Drop the derived filters (after writing parameters in a way torch cannot see)."""
        self.__dict__.pop("_ldm_cache", None)

    def _active_dropout(self) -> float:
        """Largest dropout probability of this module and its children (constructor arguments; unet.py:707, :636-639, :579)."""
        p = self.__dict__.get("_ldm_dropout_p")
        if p is None:
            p = max([float(getattr(m, "dropout_p", 0.0)) for m in self.modules()] + [0.0])
            self.__dict__["_ldm_dropout_p"] = p
        return p

    def _check_call(self, x: torch.Tensor, *others) -> bool:
        """Validates the call; True when autograd is recording and the training form must run."""
        # nn.Dropout(p) is the identity in eval mode, so a model BUILT with dropout > 0 samples correctly here; in training
        # mode the reference draws masks (also under no_grad): there is no kernel for that, and ignoring it would be silently
        # wrong, so the call is refused (the reference's trainer and every default use 0.0).
        if self.training and self._active_dropout() > 0.0:
            raise NotImplementedError("ldm_amd: dropout > 0 is only supported in eval mode (module.eval()): the training-mode "
                                      "masks of nn.Dropout have no kernel here; the reference trainer uses dropout = 0")
        if not x.is_cuda:
            raise RuntimeError("ldm_amd: tensors must live on an AMD GPU (there is no CPU fallback)")
        p = next(self.parameters(), None)
        if p is not None and p.device != x.device:
            raise RuntimeError("ldm_amd: parameters and input are on different devices")
        return torch.is_grad_enabled() and (x.requires_grad or any(torch.is_tensor(o) and o.requires_grad for o in others)
                                            or any(q.requires_grad for q in self.parameters()))


def _to_nhwc(x: torch.Tensor, dt: torch.dtype) -> torch.Tensor:
    return x.to(dt).permute(0, 2, 3, 1).contiguous()


def _to_nchw(x: torch.Tensor) -> torch.Tensor:
    return x.permute(0, 3, 1, 2).contiguous()


# ------------------------------------------------------------------------------------- kernel schedules
def run_resnet(P, n: str, r: "ResnetBlock", x, skip, rowvec):
    """unet.py:720-731 with x (and skip) NHWC; the concat [x, skip] is only ever read in place.
    rowvec: (B, Cout) view of Linear(SiLU(temb)) (bias included)."""
    g = r.norm_num_groups
    a = ops.groupnorm(x, P[_j(n, "visual.0.g")], P[_j(n, "visual.0.b")], g, a1=skip, silu=True)
    v = ops.conv_gemm(a, P[_j(n, "visual.2.w")], ksize=3, bias=P[_j(n, "visual.2.b")], rowvec=rowvec)
    a = ops.groupnorm(v, P[_j(n, "layers.0.g")], P[_j(n, "layers.0.b")], g, silu=True)
    if r.conv_shortcut is not None:
        res = ops.conv_gemm(x, P[_j(n, "conv_shortcut.w")], a1=skip, bias=P[_j(n, "conv_shortcut.b")])
    else:
        res = x
    return ops.conv_gemm(a, P[_j(n, "layers.3.w")], ksize=3, bias=P[_j(n, "layers.3.b")], residual=res)


def run_attention(P, n: str, a: "Attention", x, ctx, residual=None, ln=None, rowstats=False):
    """unet.py:641-684.  ctx None = self-attention (one fused q|k|v GEMM).
    ln = (stats, eps): x is the INPUT of the LayerNorm in front of this attention (its q / q|k|v projection runs the folded
    form); rowstats: the output projection also returns the row statistics of its result (for the next LayerNorm)."""
    C = a.inner_dim

    def proj(key):
        if ln is None:
            return ops.conv_gemm(x, P[_j(n, key + ".w")])
        return ops.conv_gemm(x, P[_j(n, key + ".w_ln")], bias=P[_j(n, key + ".b_ln")], ln=(ln[0], P[_j(n, key + ".s1")], ln[1]))

    if ctx is None and a.fused_qkv:
        qkv = proj("qkv")
        q, kv = qkv[..., :C], qkv[..., C:]
    else:
        q = proj("q_proj")
        if isinstance(ctx, _ContextKV):
            kv = ctx.of(P, n, 2 * C)
        else:
            assert ln is None or ctx is not None      # a folded self-attention always has the fused q|k|v filter
            kv = ops.conv_gemm(x if ctx is None else ctx, P[_j(n, "kv_proj.w")])
    att = ops.attention(q, kv, a.num_heads)
    return ops.conv_gemm(att, P[_j(n, "output_proj.0.w")], bias=P[_j(n, "output_proj.0.b")], residual=residual, rowstats=rowstats)


LN_EPS = 1e-5


def run_transformer_block(P, n: str, blk: "TransformerBlock", x, ctx, stats=None, rowstats=False):
    """unet.py:593-596 on (B, L, C).  stats: row statistics of x from its producer -> the LayerNorm-folded schedule (no
    LayerNorm kernel; every GEMM that writes a LayerNorm input also writes its row statistics)."""
    if stats is not None:
        x1, st1 = run_attention(P, _j(n, "attn1"), blk.attn1, x, None, residual=x, ln=(stats, LN_EPS), rowstats=True)
        x2, st2 = run_attention(P, _j(n, "attn2"), blk.attn2, x1, ctx, residual=x1, ln=(st1, LN_EPS), rowstats=True)
        k = _j(n, "feed_forward.1.w1_w2")
        y = ops.conv_gemm(x2, P[k + ".w_ln"], bias=P[k + ".b_ln"], glu=blk.activation_fn, ln=(st2, P[k + ".s1"], LN_EPS))
        return ops.conv_gemm(y, P[_j(n, "feed_forward.3.w")], bias=P[_j(n, "feed_forward.3.b")], rowstats=rowstats)
    y = ops.layernorm(x, P[_j(n, "norm1.g")], P[_j(n, "norm1.b")])
    x1 = run_attention(P, _j(n, "attn1"), blk.attn1, y, None, residual=x)
    y = ops.layernorm(x1, P[_j(n, "norm2.g")], P[_j(n, "norm2.b")])
    x2 = run_attention(P, _j(n, "attn2"), blk.attn2, y, ctx, residual=x1)
    y = ops.layernorm(x2, P[_j(n, "feed_forward.0.g")], P[_j(n, "feed_forward.0.b")])
    y = ops.conv_gemm(y, P[_j(n, "feed_forward.1.w1_w2.w")], bias=P[_j(n, "feed_forward.1.w1_w2.b")], glu=blk.activation_fn)
    return ops.conv_gemm(y, P[_j(n, "feed_forward.3.w")], bias=P[_j(n, "feed_forward.3.b")])  # no residual (:595)


def run_transformer_model(P, n: str, t: "TransformerModel", h, ctx):
    """unet.py:507-521 on the NHWC activation viewed as (B, HW, C)."""
    B, H, W, C = h.shape
    x = ops.groupnorm(h, P[_j(n, "norm.g")], P[_j(n, "norm.b")], t.norm_num_groups)
    fold = bool(P.get(_j(n, "ln_fold"))) and x.dtype == torch.bfloat16 and B * H * W > 0 and _LN_FOLD
    nb = len(t.transformer_blocks)
    if fold:
        x, stats = ops.conv_gemm(x.view(B, H * W, C), P[_j(n, "proj_in.w")], bias=P[_j(n, "proj_in.b")], rowstats=True)
        for k, blk in enumerate(t.transformer_blocks):
            x = run_transformer_block(P, _j(n, f"transformer_blocks.{k}"), blk, x, ctx, stats=stats, rowstats=k + 1 < nb)
            if k + 1 < nb:
                x, stats = x
    else:
        x = ops.conv_gemm(x.view(B, H * W, C), P[_j(n, "proj_in.w")], bias=P[_j(n, "proj_in.b")])
        for k, blk in enumerate(t.transformer_blocks):
            x = run_transformer_block(P, _j(n, f"transformer_blocks.{k}"), blk, x, ctx)
    x = ops.conv_gemm(x, P[_j(n, "proj_out.w")], bias=P[_j(n, "proj_out.b")])  # no skip connection (:521)
    return x.view(B, H, W, C)


# -------------------------------------------------------------------------------------- public modules
class ResnetBlock(_HipModule):
    """unet.py:687-731.  forward(input_tensor NCHW, temb (B, temb_channels)) -> NCHW."""

    def __init__(self, in_channels: int, out_channels: Optional[int] = None, temb_channels: int = 512,
                 norm_num_groups: int = 32, dropout: float = 0.0):
        super().__init__()
        out_channels = in_channels if out_channels is None else out_channels
        self.dropout_p = float(dropout)      # layers.2 is nn.Dropout(dropout) in the reference (:707); see _check_call
        self.visual = _seq(nn.GroupNorm(norm_num_groups, in_channels), nn.Identity(),
                           nn.Conv2d(in_channels, out_channels, 3, padding=1))
        self.time = _seq(nn.Identity(), nn.Linear(temb_channels, out_channels))
        self.layers = _seq(nn.GroupNorm(norm_num_groups, out_channels), nn.Identity(), nn.Identity(),
                           nn.Conv2d(out_channels, out_channels, 3, padding=1))
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None
        self.in_channels, self.out_channels, self.norm_num_groups = in_channels, out_channels, int(norm_num_groups)

    def forward(self, input_tensor: torch.Tensor, temb: torch.Tensor) -> torch.Tensor:
        dt = compute_dtype_of(input_tensor)
        if self._check_call(input_tensor, temb):
            return _train.apply(self, lambda ins, needs: self._train_form(dt, needs, *ins), (input_tensor, temb))
        P = self._packed_for(dt)
        rowvec = ops.conv_gemm(ops.silu(temb.to(dt)), P["time.1.w"], bias=P["time.1.b"])
        return _to_nchw(run_resnet(P, "", self, _to_nhwc(input_tensor, dt), None, rowvec))

    def _train_form(self, dt, needs, input_tensor, temb):
        T = _train.Tape(self._packed_for(dt, train=True), dt)
        te = temb.to(dt).contiguous()
        rowvec, rv_bwd = _train.conv_train(T, "time.1", ops.silu(te))
        out, b = _train.resnet_train(T, "", self, _to_nhwc(input_tensor, dt), None, rowvec)

        def bwd(dout):
            dx, _, drow = b(_to_nhwc(dout, dt))
            dtemb = ops.silu_backward(te, rv_bwd(drow.to(dt)))
            return _to_nchw(dx), dtemb
        return _to_nchw(out), bwd, T


class Attention(_HipModule):
    """unet.py:599-684.  forward(in_features (B, L, C), cross_features=None) -> (B, L, inner_dim)."""

    def __init__(self, query_dim: int, num_heads: int = 8, dim_head: int = 64, dropout: float = 0.0,
                 cross_attention_dim: Optional[int] = None):
        super().__init__()
        self.query_dim, self.num_heads, self.dim_head, self.dropout = query_dim, num_heads, dim_head, dropout
        self.dropout_p = float(dropout)      # output_proj.1 is nn.Dropout(dropout) in the reference (:636-639)
        self.cross_attention_dim = cross_attention_dim if cross_attention_dim is not None else query_dim
        self.inner_dim = dim_head * num_heads
        self.q_proj = nn.Linear(query_dim, self.inner_dim, bias=False)
        self.kv_proj = nn.Linear(self.cross_attention_dim, self.inner_dim * 2, bias=False)
        self.output_proj = _seq(nn.Linear(self.inner_dim, self.inner_dim), nn.Identity())
        self.fused_qkv = self.cross_attention_dim == query_dim

    def self_attention_only(self):
        """Used by TransformerBlock for attn1: only the fused q|k|v filter is packed."""
        self.q_proj._ldm_skip = True
        self.kv_proj._ldm_skip = True
        return self

    def forward(self, in_features: torch.Tensor, cross_features: Optional[torch.Tensor] = None) -> torch.Tensor:
        dt = compute_dtype_of(in_features)
        if self._check_call(in_features, cross_features):
            return _train.apply(self, lambda ins, needs: self._train_form(dt, needs, *ins), (in_features, cross_features))
        P = self._packed_for(dt)
        ctx = None if cross_features is None else cross_features.to(dt).contiguous()
        return run_attention(P, "", self, in_features.to(dt).contiguous(), ctx)

    def _train_form(self, dt, needs, in_features, cross_features):
        T = _train.Tape(self._packed_for(dt, train=True), dt, need_ctx_grad=needs[1])
        ctx = None if cross_features is None else cross_features.to(dt).contiguous()
        out, b = _train.attention_train(T, "", self, in_features.to(dt).contiguous(), ctx)
        return out, (lambda dout: (b(dout.to(dt)), T.dctx)), T


class _GatedProjection(nn.Module):
    """Parameter holder for GEGLU / SwiGLU (unet.py:734-790): one Linear(dim, 2*d_ff) named w1_w2."""

    def __init__(self, dim: int, d_ff: int, bias: bool = True):
        super().__init__()
        self.w1_w2 = nn.Linear(dim, 2 * d_ff, bias=bias)


class GEGLU(_GatedProjection):
    pass


class SwiGLU(_GatedProjection):
    pass


class TransformerBlock(_HipModule):
    """unet.py:524-596.  forward(hidden_states (B, L, C), encoder_hidden_states (B, S, Dc)) -> (B, L, C)."""

    def __init__(self, in_channels: int, num_attention_heads: int, cross_attention_dim: int, dropout=0.0,
                 activation_fn: str = "geglu", feed_forward_mult: int = 4):
        super().__init__()
        assert in_channels % num_attention_heads == 0, "in_channels must be divisible by num_attention_heads"
        self.dropout_p = float(dropout)      # feed_forward.2 is nn.Dropout(dropout) in the reference (:579)
        self.attention_head_dim = in_channels // num_attention_heads
        self.norm1 = nn.LayerNorm(in_channels)
        self.norm2 = nn.LayerNorm(in_channels)
        self.attn1 = Attention(in_channels, num_attention_heads, self.attention_head_dim, dropout).self_attention_only()
        # attn2 is built without num_heads in the reference: always 8 heads (unet.py:559-564, default :619)
        self.attn2 = Attention(in_channels, dim_head=self.attention_head_dim, dropout=dropout,
                               cross_attention_dim=cross_attention_dim)
        ff_dim = int(feed_forward_mult * in_channels)
        if activation_fn == "geglu":
            act = GEGLU(in_channels, ff_dim)
        elif activation_fn == "swiglu":
            act = SwiGLU(in_channels, ff_dim)
        else:
            raise ValueError(f"Unsupported activation function: {activation_fn}")
        self.activation_fn = activation_fn
        self.feed_forward = _seq(nn.LayerNorm(in_channels), act, nn.Identity(), nn.Linear(ff_dim, in_channels))

    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: torch.Tensor) -> torch.Tensor:
        dt = compute_dtype_of(hidden_states)
        if self._check_call(hidden_states, encoder_hidden_states):
            return _train.apply(self, lambda ins, needs: self._train_form(dt, needs, *ins), (hidden_states, encoder_hidden_states))
        P = self._packed_for(dt)
        return run_transformer_block(P, "", self, hidden_states.to(dt).contiguous(),
                                     encoder_hidden_states.to(dt).contiguous())

    def _train_form(self, dt, needs, hidden_states, encoder_hidden_states):
        T = _train.Tape(self._packed_for(dt, train=True), dt, need_ctx_grad=needs[1])
        out, b = _train.transformer_block_train(T, "", self, hidden_states.to(dt).contiguous(),
                                                encoder_hidden_states.to(dt).contiguous())
        return out, (lambda dout: (b(dout.to(dt)), T.dctx)), T


class TransformerModel(_HipModule):
    """unet.py:451-521.  forward(hidden_states NCHW, encoder_hidden_states (B, S, Dc)) -> NCHW."""

    def __init__(self, in_channels: int = 320, num_attention_heads: int = 16, num_layers: int = 1,
                 cross_attention_dim: int = 768, dropout: float = 0.0, norm_num_groups: int = 32,
                 feed_forward_mult: int = 4, activation_fn: str = "geglu"):
        super().__init__()
        assert in_channels % num_attention_heads == 0, "in_channels must be divisible by num_attention_heads"
        self.in_channels, self.norm_num_groups = in_channels, int(norm_num_groups)
        self.norm = nn.GroupNorm(norm_num_groups, in_channels)
        self.proj_in = nn.Conv2d(in_channels, in_channels, 1)
        self.transformer_blocks = nn.ModuleList([
            TransformerBlock(in_channels, num_attention_heads, cross_attention_dim, dropout, activation_fn, feed_forward_mult)
            for _ in range(num_layers)])
        self.proj_out = nn.Conv2d(in_channels, in_channels, 1)

    def forward(self, hidden_states: torch.Tensor, encoder_hidden_states: Optional[torch.Tensor] = None) -> torch.Tensor:
        dt = compute_dtype_of(hidden_states)
        if self._check_call(hidden_states, encoder_hidden_states):
            return _train.apply(self, lambda ins, needs: self._train_form(dt, needs, *ins), (hidden_states, encoder_hidden_states))
        P = self._packed_for(dt)
        return _to_nchw(run_transformer_model(P, "", self, _to_nhwc(hidden_states, dt),
                                              encoder_hidden_states.to(dt).contiguous()))

    def _train_form(self, dt, needs, hidden_states, encoder_hidden_states):
        T = _train.Tape(self._packed_for(dt, train=True), dt, need_ctx_grad=needs[1])
        out, b = _train.transformer_model_train(T, "", self, _to_nhwc(hidden_states, dt),
                                                encoder_hidden_states.to(dt).contiguous())
        return _to_nchw(out), (lambda dout: (_to_nchw(b(_to_nhwc(dout, dt))), T.dctx)), T


def get_time_embed(sample: torch.Tensor, timestep: Union[torch.Tensor, float, int], num_channels: int) -> torch.Tensor:
    """unet.py:159-174: sinusoidal embedding (cos first, frequency divisor num_channels/2 - 1) of a python int / float, a
    0-d tensor or a (B,) tensor, expanded to the batch of ``sample`` and cast to its dtype.  (B, num_channels)."""
    t = timestep
    if not torch.is_tensor(t):
        t = torch.tensor([t], dtype=torch.float64 if isinstance(t, float) else torch.int64)
    elif t.dim() == 0:
        t = t[None]
    t = t.to(sample.device, non_blocking=True).expand(sample.shape[0]).to(torch.float32)
    if sample.dtype in (torch.float32, torch.bfloat16):
        return ops.timestep_embedding(t, num_channels, sample.dtype)
    return ops.timestep_embedding(t, num_channels, torch.float32).to(sample.dtype)


class _Stage(_HipModule):
    """One down / mid / up stage of the UNet: the reference's block classes (unet.py:177-448) hold ``resnets``,
    ``attentions`` and a ``downsample`` / ``upsample`` conv under exactly these names.  Inside ``UNet.forward`` the stages
    are walked by the fused schedule (``_infer_steps`` / ``_train.unet_train``); called on their own (the reference's
    ``forward`` signatures, NCHW tensors in and out) they run the same kernels block by block.  Stand-alone calls are
    inference-only: gradients flow through ``UNet`` (or through ResnetBlock / TransformerModel called directly)."""

    def _begin(self, hidden_states, temb, *others):
        if self._check_call(hidden_states, temb, *others):
            raise NotImplementedError("ldm_amd: a UNet stage called on its own is inference-only; wrap the call in "
                                      "torch.no_grad() or differentiate through UNet / ResnetBlock / TransformerModel")
        dt = compute_dtype_of(hidden_states)
        P = self._packed_for(dt)
        e = None if temb is None else ops.silu(temb.to(dt))     # nn.SiLU in front of every ResnetBlock.time (unet.py:703)
        return dt, P, e

    def _resnet(self, P, j, h, skip, e):
        n = f"resnets.{j}"
        rowvec = ops.conv_gemm(e, P[n + ".time.1.w"], bias=P[n + ".time.1.b"])
        return run_resnet(P, n, self.resnets[j], h, skip, rowvec)

    def _attn(self, P, j, h, ctx):
        return run_transformer_model(P, f"attentions.{j}", self.attentions[j], h, ctx)


def _resnets(chans, temb_channels, groups, dropout):
    return nn.ModuleList([ResnetBlock(ci, co, temb_channels, groups, dropout) for ci, co in chans])


def _transformers(n, c, heads, layers, cross_dim, dropout, groups, ff_mult, act):
    return nn.ModuleList([TransformerModel(c, heads, layers, cross_dim, dropout, groups, ff_mult, act) for _ in range(n)])


class CrossAttnDownBlock(_Stage):
    """unet.py:177-237.  forward -> (hidden_states, (after layer 0, ..., after the stride-2 conv))."""

    def __init__(self, in_channels: int, out_channels: int, temb_channels: int, dropout: float = 0.0, num_layers: int = 2,
                 transformer_layers_per_block: int = 1, norm_num_groups: int = 32, num_attention_heads: int = 8,
                 cross_attention_dim: int = 1280, feed_forward_mult: int = 4, activation_fn: str = "geglu"):
        super().__init__()
        self.resnets = _resnets([(in_channels if i == 0 else out_channels, out_channels) for i in range(num_layers)],
                                temb_channels, norm_num_groups, dropout)
        self.attentions = _transformers(num_layers, out_channels, num_attention_heads, transformer_layers_per_block,
                                        cross_attention_dim, dropout, norm_num_groups, feed_forward_mult, activation_fn)
        self.downsample = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=2, padding=1, bias=True)

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None):
        dt, P, e = self._begin(hidden_states, temb, encoder_hidden_states)
        ctx = encoder_hidden_states.to(dt).contiguous()
        h = _to_nhwc(hidden_states, dt)
        states = ()
        for j in range(len(self.resnets)):
            h = self._attn(P, j, self._resnet(P, j, h, None, e), ctx)
            states += (_to_nchw(h),)
        h = ops.conv_gemm(h, P["downsample.w"], ksize=3, stride=2, bias=P["downsample.b"])
        out = _to_nchw(h)
        return out, states + (out,)


class DownBlock(_Stage):
    """unet.py:240-273."""

    def __init__(self, in_channels: int, out_channels: int, temb_channels: int, dropout: float = 0.0, num_layers: int = 1,
                 norm_num_groups: int = 32):
        super().__init__()
        self.resnets = _resnets([(in_channels if i == 0 else out_channels, out_channels) for i in range(num_layers)],
                                temb_channels, norm_num_groups, dropout)

    def forward(self, hidden_states, temb=None):
        dt, P, e = self._begin(hidden_states, temb)
        h = _to_nhwc(hidden_states, dt)
        states = ()
        for j in range(len(self.resnets)):
            h = self._resnet(P, j, h, None, e)
            states += (_to_nchw(h),)
        return states[-1], states


def _up_resnet_channels(in_channels, out_channels, final_skip_channels, num_layers):
    return [((in_channels if i == 0 else out_channels) + (final_skip_channels if i == num_layers - 1 else out_channels),
             out_channels) for i in range(num_layers)]


class UpBlock(_Stage):
    """unet.py:276-318: layer i reads cat([hidden, res_hidden_states_tuple[-i]]) -- here without materialising the concat --
    then nearest-2x upsampling folded into the 3x3 conv's gather."""

    def __init__(self, in_channels: int, out_channels: int, final_skip_channels: int, temb_channels: int, dropout: float = 0.0,
                 num_layers: int = 1, norm_num_groups: int = 32):
        super().__init__()
        self.resnets = _resnets(_up_resnet_channels(in_channels, out_channels, final_skip_channels, num_layers),
                                temb_channels, norm_num_groups, dropout)
        self.upsample = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=True)

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None):
        dt, P, e = self._begin(hidden_states, temb, *res_hidden_states_tuple)
        h = _to_nhwc(hidden_states, dt)
        for j in range(len(self.resnets)):
            h = self._resnet(P, j, h, _to_nhwc(res_hidden_states_tuple[-(j + 1)], dt), e)
        return _to_nchw(ops.conv_gemm(h, P["upsample.w"], ksize=3, upsample=True, bias=P["upsample.b"]))


class CrossAttnUpBlock(_Stage):
    """unet.py:321-387."""

    def __init__(self, in_channels: int, out_channels: int, final_skip_channels: int, temb_channels: int, dropout: float = 0.0,
                 num_layers: int = 1, transformer_layers_per_block: int = 1, norm_num_groups: int = 32,
                 num_attention_heads: int = 8, cross_attention_dim: int = 1280, feed_forward_mult: int = 4,
                 activation_fn: str = "geglu", add_upsample: bool = True):
        super().__init__()
        self.resnets = _resnets(_up_resnet_channels(in_channels, out_channels, final_skip_channels, num_layers),
                                temb_channels, norm_num_groups, dropout)
        self.attentions = _transformers(num_layers, out_channels, num_attention_heads, transformer_layers_per_block,
                                        cross_attention_dim, dropout, norm_num_groups, feed_forward_mult, activation_fn)
        self.upsample = nn.Conv2d(out_channels, out_channels, kernel_size=3, stride=1, padding=1, bias=True) if add_upsample else None

    def forward(self, hidden_states, res_hidden_states_tuple, temb=None, encoder_hidden_states=None):
        dt, P, e = self._begin(hidden_states, temb, encoder_hidden_states, *res_hidden_states_tuple)
        ctx = encoder_hidden_states.to(dt).contiguous()
        h = _to_nhwc(hidden_states, dt)
        for j in range(len(self.resnets)):
            h = self._attn(P, j, self._resnet(P, j, h, _to_nhwc(res_hidden_states_tuple[-(j + 1)], dt), e), ctx)
        if self.upsample is not None:
            h = ops.conv_gemm(h, P["upsample.w"], ksize=3, upsample=True, bias=P["upsample.b"])
        return _to_nchw(h)


class CrossAttnBlock(_Stage):
    """unet.py:390-448 (the mid block): resnet, then (transformer, resnet) pairs."""

    def __init__(self, in_channels: int, out_channels: int, temb_channels: int, dropout: float = 0.0, num_layers: int = 1,
                 transformer_layers_per_block: int = 1, norm_num_groups: int = 32, num_attention_heads: int = 8,
                 cross_attention_dim: int = 1280, feed_forward_mult: int = 4, activation_fn: str = "geglu"):
        super().__init__()
        self.resnets = _resnets([(in_channels if i == 0 else out_channels, out_channels) for i in range(num_layers + 1)],
                                temb_channels, norm_num_groups, dropout)
        self.attentions = _transformers(num_layers, out_channels, num_attention_heads, transformer_layers_per_block,
                                        cross_attention_dim, dropout, norm_num_groups, feed_forward_mult, activation_fn)

    def forward(self, hidden_states, temb=None, encoder_hidden_states=None):
        dt, P, e = self._begin(hidden_states, temb, encoder_hidden_states)
        ctx = encoder_hidden_states.to(dt).contiguous()
        h = self._resnet(P, 0, _to_nhwc(hidden_states, dt), None, e)
        for j in range(len(self.attentions)):
            h = self._resnet(P, j + 1, self._attn(P, j, h, ctx), None, e)
        return _to_nchw(h)


_DONE = object()


class _Schedule:
    """Iterator over a generator-style kernel schedule that keeps the generator's return value in `.out`."""

    def __init__(self, gen):
        self._gen = gen
        self.out = None

    def __iter__(self):
        return self

    def __next__(self):
        try:
            return next(self._gen)
        except StopIteration as stop:
            if stop.value is not None:
                self.out = stop.value
            raise StopIteration from None


class UNet(_HipModule):
    def __init__(self, num_attention_heads: int = 8, block_out_channels: List[int] = [320, 640, 1280, 1280],
                 cross_attention_dim: int = 768, in_channels: int = 4, layers_per_block: int = 2,
                 transformer_layers_per_block: int = 1, norm_num_groups: float = 32, dropout: float = 0.0,
                 feed_forward_mult: int = 4, activation_fn: str = "geglu"):
        super().__init__()
        assert len(block_out_channels) > 1  # unet.py:24
        ch = list(block_out_channels)
        self.block_out_channels = block_out_channels
        self.norm_num_groups = g = int(norm_num_groups)
        temb = ch[0] * 4
        n = len(ch)

        attn_kw = dict(transformer_layers_per_block=transformer_layers_per_block, norm_num_groups=g,
                       num_attention_heads=num_attention_heads, cross_attention_dim=cross_attention_dim,
                       feed_forward_mult=feed_forward_mult, activation_fn=activation_fn)
        self.conv_in = nn.Conv2d(in_channels, ch[0], 3, padding=1)
        self.conv_in._ldm_raw = True
        self.time_embedding = _seq(nn.Linear(ch[0], temb), nn.Identity(), nn.Linear(temb, temb))
        downs = []
        for i in range(n):      # unet.py:35-62: the last down stage has neither attention nor a downsampler
            cin = ch[0] if i == 0 else ch[i - 1]
            if i == n - 1:
                downs.append(DownBlock(cin, ch[i], temb, dropout, layers_per_block, g))
            else:
                downs.append(CrossAttnDownBlock(cin, ch[i], temb, dropout, layers_per_block, **attn_kw))
        self.down_blocks = nn.ModuleList(downs)
        self.mid_block = CrossAttnBlock(ch[-1], ch[-1], temb, dropout, 1, **attn_kw)      # unet.py:63-75
        ups = []
        for i in range(n):      # unet.py:77-111
            rev = n - 1 - i
            cin = ch[-1] if i == 0 else ch[rev + 1]
            last_skip = ch[0] if rev == 0 else ch[rev - 1]
            if i == 0:
                ups.append(UpBlock(cin, ch[rev], last_skip, temb, dropout, layers_per_block + 1, g))
            else:
                ups.append(CrossAttnUpBlock(cin, ch[rev], last_skip, temb, dropout, layers_per_block + 1,
                                            add_upsample=i < n - 1, **attn_kw))
        self.up_blocks = nn.ModuleList(ups)
        self.final_conv = _seq(nn.GroupNorm(g, ch[0]), nn.Identity(), nn.Conv2d(ch[0], in_channels, 3, padding=1))
        self.final_conv[2]._ldm_raw = True
        for m in self.modules():  # the per-block time projections run as one fused GEMM
            if isinstance(m, ResnetBlock):
                m.time[1]._ldm_skip_in_unet = True

    def forward(self, sample: torch.Tensor, timestep: Union[torch.Tensor, float, int],
                encoder_hidden_states: torch.Tensor) -> torch.Tensor:
        train = self._check_call(sample, encoder_hidden_states)
        dt = compute_dtype_of(sample)
        B = sample.shape[0]
        dev = sample.device

        # timestep forms of unet.py:159-170 (python int -> int64, float -> float64, 0-d tensor, (B,) tensor)
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.float64 if isinstance(t, float) else torch.int64)
        elif t.dim() == 0:
            t = t[None]
        t = t.to(dev, non_blocking=True).expand(B).to(torch.float32)
        if train:
            return _train.apply(self, lambda ins, needs: self._train_form(dt, t, needs, *ins), (sample, encoder_hidden_states))
        P = self._packed_for(dt)
        nmb = self._micro_batches(B)
        if nmb == 1:
            for _ in (g := self._infer(P, dt, sample, t, encoder_hidden_states)):
                pass
            return g.out
        # Opt-in: micro-batches on concurrent HIP streams.  Samples are independent through the whole net (GroupNorm, LayerNorm
        # and attention are per sample), and ~500 dependent kernels per forward each end in a drain + cross-XCD cache
        # write-back during which the chip idles (5-7 % of a forward in the kernel trace): with two half-batches in
        # flight one stream's kernels fill the other's boundaries.  The schedules advance block by block in turn so
        # that both queues stay fed; every stream has its own workspaces and allocator pool; the caller's stream
        # joins at the end.
        main = torch.cuda.current_stream(dev)
        streams = self._streams(dev, nmb)
        bounds = [B * k // nmb for k in range(nmb + 1)]
        gens = []
        for k in range(nmb):
            streams[k].wait_stream(main)
            lo, hi = bounds[k], bounds[k + 1]
            gens.append(self._infer(P, dt, sample[lo:hi], t[lo:hi], encoder_hidden_states[lo:hi]))
        live = list(range(nmb))
        while live:
            for k in list(live):
                with torch.cuda.stream(streams[k]):
                    if next(gens[k], _DONE) is _DONE:
                        live.remove(k)
        for k in range(nmb):
            main.wait_stream(streams[k])
            gens[k].out.record_stream(main)
        return torch.cat([g.out for g in gens], dim=0)

    def _micro_batches(self, B: int) -> int:
        """Concurrent micro-batches of the inference forward: opt-in through the attribute `micro_batch_streams` or
        LDM_MICRO_BATCH_STREAMS (default 1 = off), used from 16 samples up; never under stream capture (ldm/graph.py)
        and never in the training form.  Measured on configs[1] (B = 32): 2 streams +2.8 %, 3 streams -4 %, 4 -15 %
        (half-batches tile the chip less well, which eats most of what the overlapped kernel boundaries give back)."""
        n = getattr(self, "micro_batch_streams", None)
        if n is None:
            n = int(os.environ.get("LDM_MICRO_BATCH_STREAMS", "1"))
        if n <= 1 or B < 16 or B < 2 * n:
            return 1
        if torch.cuda.is_current_stream_capturing() and not getattr(self, "_ldm_allow_capture_streams", False):
            return 1
        return n

    def _streams(self, dev, n: int):
        pool = self.__dict__.setdefault("_ldm_streams", {})
        key = (dev.index, n)
        if key not in pool:
            pool[key] = [torch.cuda.Stream(device=dev) for _ in range(n)]
        return pool[key]

    def _infer(self, P, dt, sample, t, encoder_hidden_states):
        return _Schedule(self._infer_steps(P, dt, sample, t, encoder_hidden_states))

    def _infer_steps(self, P, dt, sample, t, encoder_hidden_states):
        """The inference schedule (unet.py:118-156) as a generator that yields between blocks; the value it returns
        is the output."""
        emb = ops.timestep_embedding(t, self.block_out_channels[0], dt)
        e = ops.conv_gemm(emb, P["time_embedding.0.w"], bias=P["time_embedding.0.b"], silu=True)
        e = ops.conv_gemm(e, P["time_embedding.2.w"], bias=P["time_embedding.2.b"], silu=True)  # SiLU of ResnetBlock.time
        tproj = ops.conv_gemm(e, P["time_all.w"], bias=P["time_all.b"])

        def resnet(name, r, x, skip):
            off = P[name + ".time_off"]
            return run_resnet(P, name, r, x, skip, tproj[:, off:off + r.out_channels])

        ctx = encoder_hidden_states.to(dt).contiguous()
        if "ctx_kv_all.w" in P:
            ctx = _ContextKV(P, ctx)
        h = ops.conv_in(sample.contiguous(), P["conv_in.w"], P["conv_in.b"], dt)
        skips = [h]
        for i, blk in enumerate(self.down_blocks):
            attns = getattr(blk, "attentions", None)
            for j, r in enumerate(blk.resnets):
                h = resnet(f"down_blocks.{i}.resnets.{j}", r, h, None)
                yield
                if attns is not None:
                    h = run_transformer_model(P, f"down_blocks.{i}.attentions.{j}", attns[j], h, ctx)
                    yield
                skips.append(h)
            if getattr(blk, "downsample", None) is not None:
                nm = f"down_blocks.{i}.downsample"
                h = ops.conv_gemm(h, P[nm + ".w"], ksize=3, stride=2, bias=P[nm + ".b"])
                skips.append(h)
        mid = self.mid_block
        h = resnet("mid_block.resnets.0", mid.resnets[0], h, None)
        h = run_transformer_model(P, "mid_block.attentions.0", mid.attentions[0], h, ctx)
        h = resnet("mid_block.resnets.1", mid.resnets[1], h, None)
        yield
        for i, blk in enumerate(self.up_blocks):
            attns = getattr(blk, "attentions", None)
            for j, r in enumerate(blk.resnets):
                h = resnet(f"up_blocks.{i}.resnets.{j}", r, h, skips.pop())
                yield
                if attns is not None:
                    h = run_transformer_model(P, f"up_blocks.{i}.attentions.{j}", attns[j], h, ctx)
                    yield
            if getattr(blk, "upsample", None) is not None:
                nm = f"up_blocks.{i}.upsample"
                h = ops.conv_gemm(h, P[nm + ".w"], ksize=3, upsample=True, bias=P[nm + ".b"])
        h = ops.groupnorm(h, P["final_conv.0.g"], P["final_conv.0.b"], self.norm_num_groups, silu=True)
        return ops.conv_out(h, P["final_conv.2.w"], P["final_conv.2.b"], dt)

    def _train_form(self, dt, t, needs, sample, encoder_hidden_states):
        T = _train.Tape(self._packed_for(dt, train=True), dt, need_ctx_grad=needs[1],
                        sink=getattr(self, "_ldm_grad_sink", None))
        out, b = _train.unet_train(T, self, sample, t, encoder_hidden_states.to(dt).contiguous(), needs[0])
        return out, (lambda dout: (b(dout), T.dctx)), T
