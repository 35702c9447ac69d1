"""Training form of the UNet schedule: forward that keeps what the backward needs, and the hand-scheduled backward.

``loss.backward()`` in the reference's train step (train_latent_diffusion.py:1105-1119) differentiates ldm/unet.py with
torch autograd, one node per torch op.  Here every block's forward returns its output together with a closure that maps
the output gradient to the input gradients and deposits the parameter gradients (float32, checkpoint layout) in
``Tape.grads`` under the reference's ``state_dict`` names; the closures chain exactly as the forward did.  All arithmetic
is C-ABI kernels (ops.py): the weight gradient of every conv / linear on the matrix cores (ldm_conv_wgrad), the input
gradient through the forward implicit-GEMM kernel with transposed / rotated filters (ldm_conv_gemm), flash attention
backward, GroupNorm / LayerNorm / GEGLU / SiLU backward, column sums for biases and the time-embedding add.  Gradients
that meet (residuals, skip connections) are added inside the consuming kernel where it has an ``add`` / ``residual``
input, else by ldm_add.  ``torch`` only allocates, slices and reshapes.

One ``torch.autograd.Function`` (``UNetTrainFunction`` / ``BlockTrainFunction``) wraps a whole module call, so autograd
sees ordinary ``nn.Parameter`` leaves with ``.grad`` (what ``accelerator.backward`` / DDP hooks / ``clip_grad_norm_``
/ AdamW at train_latent_diffusion.py:1119-1126 consume).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import os

import torch

from . import ops


def _j(prefix: str, name: str) -> str:
    return f"{prefix}.{name}" if prefix else name


class Tape:
    """Packed weights + the parameter gradients collected by one backward pass."""

    def __init__(self, P: dict, dtype: torch.dtype, need_ctx_grad: bool = False, sink=None):
        self.P = P
        self.dt = dtype
        self.grads: Dict[str, torch.Tensor] = {}
        self.need_ctx_grad = need_ctx_grad
        self.dctx: Optional[torch.Tensor] = None
        self.sink = sink            # ldm.parallel.GradReducer: finished parameter gradients go to its buckets at once

    # ------------------------------------------------------------------ parameter gradients
    def put(self, key: str, g: torch.Tensor):
        """A finished gradient.  Keys that are not parameter names (fused filters: qkv, time_all) are intermediate:
        their owners split them and put the parts under the parameters' names."""
        if self.sink is not None and key in self.sink.params and self.sink.params[key].requires_grad:
            g = self.sink.put(key, g)           # -> view of a flat bucket, averaged once the pass finishes
        if key in self.grads:
            g = self.grads[key] + g.reshape(self.grads[key].shape)
        self.grads[key] = g

    def weight_grad(self, name: str, a0, dy, *, a1=None, ksize=1, stride=1, upsample=False, in_place=True):
        """Weight gradient of layer ``name``; with a gradient sink it is written straight into the parameter's slice of
        the sink's flat bucket (no copy between the kernel and the optimizer)."""
        out = None
        if in_place and self.sink is not None:
            c = a0.shape[-1] + (0 if a1 is None else a1.shape[-1])
            shape = (dy.shape[-1], c) if (ksize == 1 and not upsample and stride == 1) else (dy.shape[-1], c, ksize, ksize)
            out = self.sink.dest(name + ".weight", shape)
        return ops.conv_wgrad(a0, dy, a1=a1, ksize=ksize, stride=stride, upsample=upsample, out=out,
                              accumulate=False)

    def bias_grad(self, dy: torch.Tensor, key: Optional[str] = None) -> torch.Tensor:
        """Column sums of dy; ``key`` (a parameter name): written straight into its gradient-bucket slice when there is one."""
        out = self.sink.dest(key, (1, dy.shape[-1])) if (key is not None and self.sink is not None) else None
        return ops.colsum(dy, out=out, accumulate=False).view(-1)

    def norm_dest(self, prefix: str, channels: int) -> dict:
        """dgamma / dbeta destinations of a norm layer's backward inside the gradient buckets (or nothing)."""
        if self.sink is None:
            return {}
        dg, db = self.sink.dest(prefix + ".weight", (channels,)), self.sink.dest(prefix + ".bias", (channels,))
        return {} if dg is None or db is None else dict(dgamma=dg, dbeta=db, accumulate=False)

    def add_ctx_grad(self, g: torch.Tensor):
        self.dctx = g if self.dctx is None else ops.add(self.dctx, g)


_GLU_FUSED = os.environ.get("LDM_TRAIN_GLU_FUSED", "1") != "0"      # A/B switch (tools/): 0 = plain GEMM + glu_forward pass


def glu_rows_to_checkpoint(g: torch.Tensor) -> torch.Tensor:
    """Rows in ldm_pack_weight(glu_pairs) order (blocks of 16 values then their 16 gates) -> [values | gates].
    Returns a strided view whose element order is the checkpoint's: whoever stores it (Tape.put -> the gradient bucket,
    or the reshape at the autograd boundary) makes the one copy."""
    n = g.shape[0]
    return g.view(n // 32, 2, 16, *g.shape[1:]).transpose(0, 1)


def glu_rows_to_packed(w: torch.Tensor) -> torch.Tensor:
    n = w.shape[0]
    return w.view(2, n // 32, 16, *w.shape[1:]).transpose(0, 1).reshape(w.shape)


# ------------------------------------------------------------------------------------------------ layers
def conv_train(T: Tape, name: str, a0, *, a1=None, ksize=1, stride=1, upsample=False, rowvec=None, residual=None,
               has_bias=True, grad_map=None, glu=None):
    """ops.conv_gemm + backward closure.  bwd(dy, need_dx=True, add=None) -> dx over all c0+c1 channels (+ add).
    The gradient of ``residual`` is dy itself; the gradient of ``rowvec`` is colsum(dy, segments=B) (callers do both).
    ``glu`` ("geglu" / "swiglu", bf16): the GEMM's epilogue applies the gated activation and ALSO stores the pre-activation;
    returns ((gated, pre), bwd) where bwd takes the gradient of ``pre``."""
    P = T.P
    out = ops.conv_gemm(a0, P[name + ".w"], a1=a1, ksize=ksize, stride=stride, upsample=upsample,
                        bias=P[name + ".b"] if has_bias else None, rowvec=rowvec, residual=residual,
                        glu=glu, glu_preact=glu is not None)
    in_hw = (a0.shape[1], a0.shape[2]) if a0.dim() == 4 else None

    def bwd(dy, need_dx=True, add=None, bias_from=None):
        gw = T.weight_grad(name, a0, dy, a1=a1, ksize=ksize, stride=stride, upsample=upsample, in_place=grad_map is None)
        T.put(name + ".weight", gw if grad_map is None else grad_map(gw))
        if has_bias:
            gb = (T.bias_grad(dy, name + ".bias" if grad_map is None else None)) if bias_from is None else bias_from
            T.put(name + ".bias", gb if grad_map is None else grad_map(gb))
        if not need_dx:
            return None
        dx = ops.conv_dgrad(dy, P[name + ".wd"], ksize=ksize, stride=stride, upsample=upsample, in_hw=in_hw, residual=add)
        return dx.view(*a0.shape[:-1], dx.shape[-1]) if dx.dim() != a0.dim() else dx

    return out, bwd


def resnet_train(T: Tape, n: str, r, x, skip, rowvec):
    """unet.py:720-731.  bwd(dout) -> (dx, dskip or None, drowvec (B, Cout) float32)."""
    P, g = T.P, r.norm_num_groups
    B = x.shape[0]
    a, st1 = ops.groupnorm_with_stats(x, P[_j(n, "visual.0.g")], P[_j(n, "visual.0.b")], g, a1=skip, silu=True)
    v, v_bwd = conv_train(T, _j(n, "visual.2"), a, ksize=3, rowvec=rowvec)
    a2, st2 = ops.groupnorm_with_stats(v, P[_j(n, "layers.0.g")], P[_j(n, "layers.0.b")], g, silu=True)
    if r.conv_shortcut is not None:
        res, sc_bwd = conv_train(T, _j(n, "conv_shortcut"), x, a1=skip)
    else:
        res, sc_bwd = x, None
    out, o_bwd = conv_train(T, _j(n, "layers.3"), a2, ksize=3, residual=res)

    def bwd(dout):
        da2 = o_bwd(dout)
        dv, dg, db = ops.groupnorm_backward(v, st2, P[_j(n, "layers.0.g")], P[_j(n, "layers.0.b")], g, da2, silu=True,
                                            **T.norm_dest(_j(n, "layers.0"), v.shape[-1]))
        T.put(_j(n, "layers.0.weight"), dg)
        T.put(_j(n, "layers.0.bias"), db)
        drow = ops.colsum(dv, segments=B)                                   # (B, Cout) float32
        da = v_bwd(dv, bias_from=ops.colsum(drow).view(-1))
        dres = sc_bwd(dout) if sc_bwd is not None else dout                 # gradient arriving over the shortcut
        dxc, dg, db = ops.groupnorm_backward(x, st1, P[_j(n, "visual.0.g")], P[_j(n, "visual.0.b")], g, da, a1=skip,
                                             silu=True, add=dres, split=True,
                                             **T.norm_dest(_j(n, "visual.0"), da.shape[-1]))
        T.put(_j(n, "visual.0.weight"), dg)
        T.put(_j(n, "visual.0.bias"), db)
        if skip is None:
            return dxc, None, drow
        return dxc[0], dxc[1], drow                                          # concat gradient, split where produced

    return out, bwd


def attention_train(T: Tape, n: str, a, x, ctx, residual=None):
    """unet.py:641-684.  bwd(dout) -> dx (the gradient of ``residual`` is dout: the caller adds it).  The context
    gradient, when wanted, goes to T.add_ctx_grad."""
    P = T.P
    C = a.inner_dim
    fused = ctx is None and a.fused_qkv
    if fused:
        qkv, qkv_bwd = conv_train(T, _j(n, "qkv"), x, has_bias=False)
        q, kv = qkv[..., :C], qkv[..., C:]
    else:
        q, q_bwd = conv_train(T, _j(n, "q_proj"), x, has_bias=False)
        kv, kv_bwd = conv_train(T, _j(n, "kv_proj"), x if ctx is None else ctx, has_bias=False)
    att, lse = ops.attention_train(q, kv, a.num_heads)
    out, o_bwd = conv_train(T, _j(n, "output_proj.0"), att, residual=residual)

    def bwd(dout):
        datt = o_bwd(dout)
        if fused:
            dqkv = torch.empty_like(qkv)
            ops.attention_backward(q, kv, att, datt, lse, a.num_heads, dq=dqkv[..., :C], dkv=dqkv[..., C:])
            dx = qkv_bwd(dqkv)
            gw = T.grads.pop(_j(n, "qkv.weight"))                           # rows: q | k | v
            T.put(_j(n, "q_proj.weight"), gw[:C])
            T.put(_j(n, "kv_proj.weight"), gw[C:])
            return dx
        dq, dkv = ops.attention_backward(q, kv, att, datt, lse, a.num_heads)
        dx = q_bwd(dq)
        if ctx is None:
            return kv_bwd(dkv, add=dx)
        dc = kv_bwd(dkv, need_dx=T.need_ctx_grad)
        if dc is not None:
            T.add_ctx_grad(dc)
        return dx

    return out, bwd


def transformer_block_train(T: Tape, n: str, blk, x, ctx):
    """unet.py:593-596 on (B, L, C).  bwd(dout) -> dx."""
    P = T.P

    def ln(key, t):
        return ops.layernorm(t, P[_j(n, key + ".g")], P[_j(n, key + ".b")])

    def ln_bwd(key, t, dy, add=None):
        dx, dg, db = ops.layernorm_backward(t, P[_j(n, key + ".g")], dy, add=add, **T.norm_dest(_j(n, key), t.shape[-1]))
        T.put(_j(n, key + ".weight"), dg)
        T.put(_j(n, key + ".bias"), db)
        return dx

    y1 = ln("norm1", x)
    x1, a1_bwd = attention_train(T, _j(n, "attn1"), blk.attn1, y1, None, residual=x)
    y2 = ln("norm2", x1)
    x2, a2_bwd = attention_train(T, _j(n, "attn2"), blk.attn2, y2, ctx, residual=x1)
    y3 = ln("feed_forward.0", x2)
    if y3.dtype == torch.bfloat16 and _GLU_FUSED:      # one launch: the GEMM's GLU epilogue also stores the pre-activation the backward needs
        (gl, hpre), h_bwd = conv_train(T, _j(n, "feed_forward.1.w1_w2"), y3, grad_map=glu_rows_to_checkpoint,
                                       glu=blk.activation_fn)
    else:
        hpre, h_bwd = conv_train(T, _j(n, "feed_forward.1.w1_w2"), y3,      # packed (value|gate interleaved) columns
                                 grad_map=glu_rows_to_checkpoint)
        gl = ops.glu_forward(hpre, blk.activation_fn)
    out, o_bwd = conv_train(T, _j(n, "feed_forward.3"), gl)                 # no residual round the FF (:595)

    def bwd(dout):
        dgl = o_bwd(dout)
        dh = ops.glu_backward(hpre, dgl, blk.activation_fn)
        dy3 = h_bwd(dh)
        dx2 = ln_bwd("feed_forward.0", x2, dy3)
        dy2 = a2_bwd(dx2)
        dx1 = ln_bwd("norm2", x1, dy2, add=dx2)
        dy1 = a1_bwd(dx1)
        return ln_bwd("norm1", x, dy1, add=dx1)

    return out, bwd


def transformer_model_train(T: Tape, n: str, t, h, ctx):
    """unet.py:507-521 on the NHWC activation.  bwd(dy NHWC) -> dh NHWC."""
    P = T.P
    B, H, W, C = h.shape
    x0, st = ops.groupnorm_with_stats(h, P[_j(n, "norm.g")], P[_j(n, "norm.b")], t.norm_num_groups)
    x, in_bwd = conv_train(T, _j(n, "proj_in"), x0.view(B, H * W, C))
    blk_bwds = []
    for k, blk in enumerate(t.transformer_blocks):
        x, b = transformer_block_train(T, _j(n, f"transformer_blocks.{k}"), blk, x, ctx)
        blk_bwds.append(b)
    y, out_bwd = conv_train(T, _j(n, "proj_out"), x)                        # no skip connection (:521)

    def bwd(dy):
        dx = out_bwd(dy.reshape(B, H * W, C))
        for b in reversed(blk_bwds):
            dx = b(dx)
        dx0 = in_bwd(dx)
        dh, dg, db = ops.groupnorm_backward(h, st, P[_j(n, "norm.g")], P[_j(n, "norm.b")], t.norm_num_groups,
                                            dx0.view(B, H, W, C), **T.norm_dest(_j(n, "norm"), C))
        T.put(_j(n, "norm.weight"), dg)
        T.put(_j(n, "norm.bias"), db)
        return dh

    return y.view(B, H, W, C), bwd


# --------------------------------------------------------------------------------- first / last layers
def _pad_channels(x: torch.Tensor, c: int) -> torch.Tensor:
    """(.., c0) -> (.., c) zero-padded copy (plumbing for the 4-channel ends of the net)."""
    if x.shape[-1] == c:
        return x.contiguous()
    out = x.new_zeros((*x.shape[:-1], c))
    out[..., :x.shape[-1]] = x
    return out


def _edge_pad(dt: torch.dtype) -> int:
    return 32 if dt == torch.bfloat16 else 16      # channel granularity of the implicit-GEMM kernels


def conv_in_train(T: Tape, net, sample, need_dx: bool):
    """conv_in (unet.py:26, :135).  bwd(dh NHWC) -> dsample NCHW or None."""
    P, dt = T.P, T.dt
    h = ops.conv_in(sample.contiguous(), P["conv_in.w"], P["conv_in.b"], dt)
    cin = sample.shape[1]

    def bwd(dh):
        xs = _pad_channels(sample.to(dt).permute(0, 2, 3, 1), 8 * ((cin + 7) // 8))
        gw = ops.conv_wgrad(xs, dh, ksize=3)
        T.put("conv_in.weight", gw[:, :cin].contiguous())
        T.put("conv_in.bias", T.bias_grad(dh))
        if not need_dx:
            return None
        pc = _edge_pad(dt)
        w = net.conv_in.weight.detach().float()
        wp = w.new_zeros((w.shape[0], pc, 3, 3))
        wp[:, :cin] = w
        dx = ops.conv_gemm(dh, ops.pack_weight_dgrad(wp, dt), ksize=3)
        return dx[..., :cin].permute(0, 3, 1, 2).contiguous()

    return h, bwd


def conv_out_train(T: Tape, net, h):
    """final_conv.2 (unet.py:115, :155) on the GroupNorm+SiLU'ed activation.  bwd(dout NCHW) -> dh NHWC."""
    P, dt = T.P, T.dt
    out = ops.conv_out(h, P["final_conv.2.w"], P["final_conv.2.b"], dt)
    cout = out.shape[1]

    def bwd(dout):
        pc = _edge_pad(dt)
        dy = _pad_channels(dout.to(dt).permute(0, 2, 3, 1), pc)             # (B, H, W, pc), zero beyond cout
        gw = ops.conv_wgrad(h, dy, ksize=3)
        T.put("final_conv.2.weight", gw[:cout].contiguous())
        T.put("final_conv.2.bias", T.bias_grad(dy)[:cout].contiguous())
        w = net.final_conv[2].weight.detach().float()
        wp = w.new_zeros((pc, *w.shape[1:]))
        wp[:cout] = w
        return ops.conv_gemm(dy, ops.pack_weight_dgrad(wp, dt), ksize=3)

    return out, bwd


# ------------------------------------------------------------------------------------------- whole net
def unet_train(T: Tape, net, sample, t_f32, ctx, need_dsample: bool):
    """UNet.forward (unet.py:118-156) in training form.  Returns (noise_pred NCHW, bwd) with
    bwd(dout NCHW) -> dsample or None; parameter gradients land in T.grads, the context gradient in T.dctx."""
    P, dt = T.P, T.dt
    B = sample.shape[0]
    g = net.norm_num_groups

    emb = ops.timestep_embedding(t_f32, net.block_out_channels[0], dt)
    e0, e0_bwd = conv_train(T, "time_embedding.0", emb)
    e0s = ops.silu(e0)
    e1, e1_bwd = conv_train(T, "time_embedding.2", e0s)
    e1s = ops.silu(e1)                                                      # SiLU in front of ResnetBlock.time (:703)
    tproj, tp_bwd = conv_train(T, "time_all", e1s)

    tape: List[Callable] = []       # backward steps in forward order; each maps the running gradient dh -> dh
    drows: Dict[str, torch.Tensor] = {}
    dskips: Dict[int, torch.Tensor] = {}
    skips: List = []                # (tensor, index) in push order (unet.py:136-142)
    npushed = [0]

    def push_skip(h):
        sid = npushed[0]
        npushed[0] += 1
        skips.append((h, sid))
        tape.append(lambda dh: ops.add(dh, dskips.pop(sid)))   # the gradient that came down the up path joins here

    def resnet(name, r, x, skip_entry):
        off = P[name + ".time_off"]
        skip, sid = skip_entry if skip_entry is not None else (None, None)
        out, b = resnet_train(T, name, r, x, skip, tproj[:, off:off + r.out_channels])

        def bwd(dh):
            dx, dskip, drow = b(dh)
            drows[name] = drow
            if sid is not None:
                dskips[sid] = dskip
            return dx
        tape.append(bwd)
        return out

    def transformer(name, t, h):
        out, b = transformer_model_train(T, name, t, h, ctx)
        tape.append(b)
        return out

    def conv(name, h, **kw):
        out, b = conv_train(T, name, h, ksize=3, **kw)
        tape.append(b)
        return out

    h, cin_bwd = conv_in_train(T, net, sample, need_dsample)
    push_skip(h)
    for i, blk in enumerate(net.down_blocks):
        attns = getattr(blk, "attentions", None)
        for j, r in enumerate(blk.resnets):
            h = resnet(f"down_blocks.{i}.resnets.{j}", r, h, None)
            if attns is not None:
                h = transformer(f"down_blocks.{i}.attentions.{j}", attns[j], h)
            push_skip(h)
        if getattr(blk, "downsample", None) is not None:
            h = conv(f"down_blocks.{i}.downsample", h, stride=2)
            push_skip(h)
    mid = net.mid_block
    h = resnet("mid_block.resnets.0", mid.resnets[0], h, None)
    h = transformer("mid_block.attentions.0", mid.attentions[0], h)
    h = resnet("mid_block.resnets.1", mid.resnets[1], h, None)
    for i, blk in enumerate(net.up_blocks):
        attns = getattr(blk, "attentions", None)
        for j, r in enumerate(blk.resnets):
            h = resnet(f"up_blocks.{i}.resnets.{j}", r, h, skips.pop())
            if attns is not None:
                h = transformer(f"up_blocks.{i}.attentions.{j}", attns[j], h)
        if getattr(blk, "upsample", None) is not None:
            h = conv(f"up_blocks.{i}.upsample", h, upsample=True)
    hf = h
    a, stf = ops.groupnorm_with_stats(hf, P["final_conv.0.g"], P["final_conv.0.b"], g, silu=True)
    out, cout_bwd = conv_out_train(T, net, a)

    def bwd(dout):
        da = cout_bwd(dout)
        dh, dg, db = ops.groupnorm_backward(hf, stf, P["final_conv.0.g"], P["final_conv.0.b"], g, da, silu=True,
                                            **T.norm_dest("final_conv.0", da.shape[-1]))
        T.put("final_conv.0.weight", dg)
        T.put("final_conv.0.bias", db)
        while tape:
            dh = tape.pop()(dh)
        dsample = cin_bwd(dh)
        # time path: the per-resnet row-vector gradients, in the column order of the fused projection
        resnets = [(nm, P[nm + ".time_off"]) for nm in drows]
        resnets.sort(key=lambda e: e[1])
        dtp = torch.cat([drows[nm] for nm, _ in resnets], dim=1).to(dt)     # (B, sum Cout)
        de1s = tp_bwd(dtp)
        gw, gb = T.grads.pop("time_all.weight"), T.grads.pop("time_all.bias")
        for nm, off in resnets:
            c = drows[nm].shape[1]
            T.put(nm + ".time.1.weight", gw[off:off + c])
            T.put(nm + ".time.1.bias", gb[off:off + c])
        de1 = ops.silu_backward(e1, de1s)
        de0s = e1_bwd(de1)
        de0 = ops.silu_backward(e0, de0s)
        e0_bwd(de0, need_dx=False)
        return dsample

    return out, bwd


def _dense(t: torch.Tensor) -> torch.Tensor:
    """Gradients that are channel slices of a concat gradient are consumed by kernels that want dense rows."""
    return t if t.is_contiguous() else t.contiguous()


# ------------------------------------------------------------------------------------ autograd boundary
class TrainFunction(torch.autograd.Function):
    """One autograd node per module call.  ``build(inputs, needs) -> (out, bwd, tape)`` runs the training-form forward
    (``needs[i]``: whether input i wants a gradient);
    ``bwd(dout)`` returns the input gradients (a tuple as long as ``inputs``; None where not wanted) and leaves the
    parameter gradients in ``tape.grads`` under ``names``."""

    @staticmethod
    def forward(ctx, build, names, needs, *tensors):
        n_in = len(needs)
        inputs, params = tensors[:n_in], tensors[n_in:]
        out, bwd, tape = build(inputs, needs)
        ctx.bwd, ctx.tape, ctx.names, ctx.n_in = bwd, tape, names, n_in
        ctx.in_meta = [(t.dtype, t.shape) if torch.is_tensor(t) else None for t in inputs]
        ctx.param_meta = [(p.dtype, p.shape) for p in params]
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        if ctx.bwd is None:
            raise RuntimeError("ldm_amd: backward through this UNet call ran twice (activations are freed after the first)")
        sink = ctx.tape.sink
        if sink is not None:
            sink.begin()
        din = ctx.bwd(_dense(dout))
        grads = ctx.tape.grads
        if sink is not None:
            grads.update(sink.finish())          # all-reduces launched during the pass are complete (stream-ordered)
        res = [None, None, None]
        for i, meta in enumerate(ctx.in_meta):
            g = din[i] if din is not None and i < len(din) else None
            if g is None or meta is None or not ctx.needs_input_grad[3 + i]:
                res.append(None)
            else:
                res.append(g.to(meta[0]).reshape(meta[1]))
        for j, (name, (dtype, shape)) in enumerate(zip(ctx.names, ctx.param_meta)):
            g = grads.get(name)
            if g is None or not ctx.needs_input_grad[3 + ctx.n_in + j]:
                res.append(None)
            elif sink is not None and dtype == torch.float32 and sink.owns(name, g) and sink.params[name].grad is None:
                sink.params[name].grad = g.view(shape)     # the bucket slice IS the gradient (gradient_as_bucket_view)
                res.append(None)
            else:
                g = g.reshape(shape).to(dtype)
                res.append(g.clone() if sink is not None and g.dtype == torch.float32 else g)   # never alias a bucket
        ctx.bwd = ctx.tape = None      # drop the saved activations
        return tuple(res)


def apply(module, build, inputs):
    """Run ``build`` under TrainFunction with the module's parameters as autograd leaves."""
    named = [(n, p) for n, p in module.named_parameters()]
    names = [n for n, _ in named]
    needs = tuple(bool(torch.is_tensor(t) and t.requires_grad) for t in inputs)
    return TrainFunction.apply(build, names, needs, *inputs, *[p for _, p in named])
