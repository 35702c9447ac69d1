"""The VAE-decoder leg of ``generate`` on the MI355X kernels.

The reference ends its sampling loop with ``vae.decode(latents / scaling_factor, return_dict=False)[0]`` and a conversion to
uint8 (ldm/tools/train_latent_diffusion.py:596-601); ``vae`` is a ``diffusers.AutoencoderKL`` (third-party; layer stack drawn
in the reference's VAE.md:119-157).  ``AutoencoderKL`` below is the decode half of that module with diffusers' state-dict
keys (``post_quant_conv.*``, ``decoder.*``; ``encoder.*`` / ``quant_conv.*`` of a full checkpoint are ignored on load), so
the trainer's ``vae.decode(...)`` / ``vae.config.scaling_factor`` call sites keep working.  Inference only -- the trainer never
differentiates through the decoder.  Encoding is not built (the trainer encodes offline, train_latent_diffusion.py:1039).

Kernel schedule: everything is NHWC in the compute dtype (bf16 storage / fp32 accumulation under bf16 autocast or for bf16
inputs, exact fp32 otherwise).  Resnets are GroupNorm+SiLU -> 3x3 implicit GEMM (bias) -> GroupNorm+SiLU -> 3x3 implicit GEMM
(bias + residual, the 1x1 shortcut GEMM where channels change); the upsamplers read their input through the GEMM's nearest-2x
addressing, nothing is materialised.  The mid-block attention is one head as wide as the block (512): wider than any flash
tile, and small (L = latent pixels), so its scores are materialised per sample by a plain GEMM into float32, normalised by
ldm_softmax_rows and multiplied with V by a second GEMM -- V is produced already transposed (as W_v X^T) so that it is the
K-contiguous operand of that GEMM, and its bias is added after the product (softmax rows sum to one).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import List, Optional, Sequence

import torch
from torch import nn

from . import ops
from .unet import _HipModule, _j, compute_dtype_of

GN_EPS = 1e-6
_MAX_OPERAND_ELEMENTS = 1 << 30          # ldm_conv_gemm addresses an operand through 32-bit byte offsets (2 GiB in bf16)


class DecoderOutput:
    def __init__(self, sample: torch.Tensor):
        self.sample = sample


class _Resnet(nn.Module):
    def __init__(self, cin: int, cout: int, groups: int):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, cin, eps=GN_EPS)
        self.conv1 = nn.Conv2d(cin, cout, 3, padding=1)
        self.norm2 = nn.GroupNorm(groups, cout, eps=GN_EPS)
        self.conv2 = nn.Conv2d(cout, cout, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(cin, cout, 1) if cin != cout else None
        self.groups = groups


class _Attention(nn.Module):
    def __init__(self, channels: int, groups: int):
        super().__init__()
        self.group_norm = nn.GroupNorm(groups, channels, eps=GN_EPS)
        self.to_q = nn.Linear(channels, channels)
        self.to_k = nn.Linear(channels, channels)
        self.to_v = nn.Linear(channels, channels)
        self.to_out = nn.ModuleList([nn.Linear(channels, channels), nn.Identity()])
        self.groups, self.channels = groups, channels


class _MidBlock(nn.Module):
    def __init__(self, channels: int, groups: int):
        super().__init__()
        self.attentions = nn.ModuleList([_Attention(channels, groups)])
        self.resnets = nn.ModuleList([_Resnet(channels, channels, groups), _Resnet(channels, channels, groups)])


class _Upsampler(nn.Module):
    def __init__(self, channels: int):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)


class _UpBlock(nn.Module):
    def __init__(self, cin: int, cout: int, layers: int, groups: int, upsample: bool):
        super().__init__()
        self.resnets = nn.ModuleList([_Resnet(cin if j == 0 else cout, cout, groups) for j in range(layers)])
        if upsample:
            self.upsamplers = nn.ModuleList([_Upsampler(cout)])


class Decoder(nn.Module):
    def __init__(self, latent_channels: int, out_channels: int, block_out_channels: Sequence[int], layers_per_block: int,
                 groups: int):
        super().__init__()
        rev = list(block_out_channels)[::-1]
        self.conv_in = nn.Conv2d(latent_channels, rev[0], 3, padding=1)
        self.mid_block = _MidBlock(rev[0], groups)
        blocks: List[nn.Module] = []
        prev = rev[0]
        for i, c in enumerate(rev):
            blocks.append(_UpBlock(prev, c, layers_per_block + 1, groups, upsample=i != len(rev) - 1))
            prev = c
        self.up_blocks = nn.ModuleList(blocks)
        self.conv_norm_out = nn.GroupNorm(groups, rev[-1], eps=GN_EPS)
        self.conv_out = nn.Conv2d(rev[-1], out_channels, 3, padding=1)
        self.conv_in._ldm_raw = True       # few input / output channels: float32 OIHW filters, dedicated kernels
        self.conv_out._ldm_raw = True


# ------------------------------------------------------------------------------------- kernel schedules
def run_resnet(P, n: str, r: _Resnet, x):
    a = ops.groupnorm(x, P[_j(n, "norm1.g")], P[_j(n, "norm1.b")], r.groups, silu=True, eps=GN_EPS)
    v = ops.conv_gemm(a, P[_j(n, "conv1.w")], ksize=3, bias=P[_j(n, "conv1.b")])
    a = ops.groupnorm(v, P[_j(n, "norm2.g")], P[_j(n, "norm2.b")], r.groups, silu=True, eps=GN_EPS)
    res = x if r.conv_shortcut is None else ops.conv_gemm(x, P[_j(n, "conv_shortcut.w")], bias=P[_j(n, "conv_shortcut.b")])
    return ops.conv_gemm(a, P[_j(n, "conv2.w")], ksize=3, bias=P[_j(n, "conv2.b")], residual=res)


def run_attention(P, n: str, a: _Attention, x):
    """x: NHWC (B, H, W, C) -> same.  One head of C channels, softmax(q k^T / sqrt(C)) v, + residual."""
    B, H, W, C = x.shape
    L = H * W
    dt = x.dtype
    mult = 32 if dt == torch.bfloat16 else 16          # the GEMM's K granularity: the score matrix is the K operand of P V
    Lp = (L + mult - 1) // mult * mult
    xs = x.view(B, L, C)
    xn = ops.groupnorm(xs, P[_j(n, "group_norm.g")], P[_j(n, "group_norm.b")], a.groups, eps=GN_EPS)
    q = ops.conv_gemm(xn, P[_j(n, "to_q.w")], bias=P[_j(n, "to_q.b")])
    if Lp != L:                                        # zero rows: masked columns of the scores, zero columns of V^T
        xp = torch.zeros((B, Lp, C), dtype=dt, device=x.device)
        xp[:, :L] = xn
    else:
        xp = xn
    k = ops.conv_gemm(xp, P[_j(n, "to_k.w")], bias=P[_j(n, "to_k.b")])
    wv, bv = P[_j(n, "to_v.w")], P[_j(n, "to_v.b")]
    outs = []
    for b in range(B):
        vt = ops.conv_gemm(wv, xp[b])                                              # (C, Lp) = (X W_v^T)^T, bias deferred
        s = ops.conv_gemm(q[b], k[b], out_dtype=torch.float32)                     # (L, Lp) scores, float32
        p = ops.softmax_rows(s, L, float(C) ** -0.5, dt)
        outs.append(ops.conv_gemm(p, vt, bias=bv))                                 # rows of p sum to 1: + b_v afterwards
    o = outs[0][None] if B == 1 else torch.stack(outs, 0)
    y = ops.conv_gemm(o, P[_j(n, "to_out.0.w")], bias=P[_j(n, "to_out.0.b")], residual=xs)
    return y.view(B, H, W, C)


class AutoencoderKL(_HipModule):
    """Decode half of diffusers' AutoencoderKL (SD-1.x defaults).  ``decode(z)`` takes NCHW latents (already divided by
    ``config.scaling_factor``, train_latent_diffusion.py:596) and returns NCHW images in [-1, 1]."""

    def __init__(self, in_channels: int = 3, out_channels: int = 3, block_out_channels: Sequence[int] = (128, 256, 512, 512),
                 layers_per_block: int = 2, latent_channels: int = 4, norm_num_groups: int = 32,
                 scaling_factor: float = 0.18215):
        super().__init__()
        self.config = SimpleNamespace(in_channels=in_channels, out_channels=out_channels,
                                      block_out_channels=tuple(block_out_channels), layers_per_block=layers_per_block,
                                      latent_channels=latent_channels, norm_num_groups=norm_num_groups,
                                      scaling_factor=scaling_factor)
        self.post_quant_conv = nn.Conv2d(latent_channels, latent_channels, 1)
        self.post_quant_conv._ldm_raw = True
        self.decoder = Decoder(latent_channels, out_channels, block_out_channels, layers_per_block, norm_num_groups)

    # a full diffusers checkpoint also holds the encoder half
    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        kept = {k: v for k, v in state_dict.items() if not (k.startswith("encoder.") or k.startswith("quant_conv."))}
        return super().load_state_dict(kept, strict=strict, **kw)

    def encode(self, *_a, **_k):
        raise NotImplementedError("ldm_amd: the VAE encoder is outside the accelerated path (the trainer encodes offline)")

    def _max_elements_per_sample(self, h: int, w: int) -> int:
        rev = list(self.config.block_out_channels)[::-1]
        worst, prev, pix = h * w * rev[0], rev[0], h * w
        for i, c in enumerate(rev):
            worst = max(worst, pix * max(prev, c))
            if i != len(rev) - 1:
                pix *= 4
                worst = max(worst, pix * c)
            prev = c
        return worst

    @torch.no_grad()
    def decode(self, z: torch.Tensor, return_dict: bool = True, generator=None, micro_batch: Optional[int] = None):
        if z.dim() != 4 or z.shape[1] != self.config.latent_channels:
            raise ValueError(f"decode: expected (B, {self.config.latent_channels}, h, w) latents, got {tuple(z.shape)}")
        self._check_call(z)
        dt = compute_dtype_of(z)
        P = self._packed_for(dt)
        B, _, h, w = z.shape
        cap = max(1, (_MAX_OPERAND_ELEMENTS - 1) // self._max_elements_per_sample(h, w))
        mb = min(B, cap if micro_batch is None else max(1, min(int(micro_batch), cap)))
        parts = [self._decode_rows(P, dt, z[i:i + mb]) for i in range(0, B, mb)] if B else []
        if not parts:
            scale = 2 ** (len(self.config.block_out_channels) - 1)
            img = torch.empty((0, self.config.out_channels, h * scale, w * scale), dtype=dt, device=z.device)
        else:
            img = parts[0] if len(parts) == 1 else torch.cat(parts, 0)
        return DecoderOutput(img) if return_dict else (img,)

    def _decode_rows(self, P, dt, z):
        d = self.decoder
        zq = ops.conv1x1_nchw(z, P["post_quant_conv.w"], P["post_quant_conv.b"])
        x = ops.conv_in(zq, P["decoder.conv_in.w"], P["decoder.conv_in.b"], dt)
        mid = d.mid_block
        x = run_resnet(P, "decoder.mid_block.resnets.0", mid.resnets[0], x)
        x = run_attention(P, "decoder.mid_block.attentions.0", mid.attentions[0], x)
        x = run_resnet(P, "decoder.mid_block.resnets.1", mid.resnets[1], x)
        for i, blk in enumerate(d.up_blocks):
            for j, r in enumerate(blk.resnets):
                x = run_resnet(P, f"decoder.up_blocks.{i}.resnets.{j}", r, x)
            if hasattr(blk, "upsamplers"):
                nm = f"decoder.up_blocks.{i}.upsamplers.0.conv"
                x = ops.conv_gemm(x, P[nm + ".w"], ksize=3, upsample=True, bias=P[nm + ".b"])
        x = ops.groupnorm(x, P["decoder.conv_norm_out.g"], P["decoder.conv_norm_out.b"], self.config.norm_num_groups,
                          silu=True, eps=GN_EPS)
        return ops.conv_out(x, P["decoder.conv_out.w"], P["decoder.conv_out.b"], dt)

    forward = decode


def to_uint8(image: torch.Tensor) -> torch.Tensor:
    """``(255 * (image / 2 + 0.5).clamp(0, 1)).to(torch.uint8)`` (train_latent_diffusion.py:600-601) as one kernel."""
    return ops.image_to_uint8(image)
