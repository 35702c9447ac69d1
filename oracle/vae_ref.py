"""CPU oracle for the VAE decoder leg of ``generate`` (TEST INFRASTRUCTURE ONLY).

Only ``tests/`` (and checkers like ``__graft_entry__.smoke()``) may import this file; the product path never does.

PARITY UNPINNED.  The reference calls ``vae.decode(latents, return_dict=False)[0]`` (ldm/tools/train_latent_diffusion.py:597)
on a ``diffusers.AutoencoderKL``: third-party code, version unpinned in the reference (pyproject.toml:28), absent in this
offline image, and the reference holds no fixture of its outputs.  What is restated here is the published architecture of
``AutoencoderKL.decode`` / ``Decoder`` / ``UNetMidBlock2D`` / ``UpDecoderBlock2D`` / ``ResnetBlock2D`` / ``Attention`` /
``Upsample2D`` as the reference's own VAE.md:119-157 draws it (post_quant_conv 1x1 -> conv 3x3 -> mid: resnet, single-head
self-attention, resnet -> 4 up blocks x 3 resnets, nearest-2x + conv on the first three -> GroupNorm -> SiLU -> conv 3x3),
with diffusers' constants: GroupNorm eps 1e-6 and 32 groups everywhere, resnets without time embedding and with a 1x1
``conv_shortcut`` when the channel count changes, attention = GroupNorm -> biased to_q/to_k/to_v -> softmax(q k^T / sqrt(C)) v ->
biased to_out.0 -> + residual, output scale factors 1.  State-dict keys are diffusers' (``post_quant_conv.*``,
``decoder.conv_in.*``, ``decoder.mid_block.resnets.N.*``, ``decoder.mid_block.attentions.0.*``,
``decoder.up_blocks.N.resnets.M.*``, ``decoder.up_blocks.N.upsamplers.0.conv.*``, ``decoder.conv_norm_out.*``,
``decoder.conv_out.*``).  Every op is a plain torch.nn.functional call, evaluated in the dtype of ``z`` (float64 in the tests).
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

from .unet_ref import fill_state_dict  # noqa: F401  (same deterministic filler as the UNet fixtures)

SD = Dict[str, torch.Tensor]
GN_EPS = 1e-6


class VAEDecoderConfig:
    """The decoder-side constructor arguments of diffusers' AutoencoderKL (SD-1.x values as defaults; VAE.md:76-83)."""

    def __init__(self, latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2,
                 norm_num_groups=32, scaling_factor=0.18215):
        self.latent = int(latent_channels)
        self.out_ch = int(out_channels)
        self.chans = [int(c) for c in block_out_channels]
        self.layers = int(layers_per_block)
        self.groups = int(norm_num_groups)
        self.scaling_factor = float(scaling_factor)


def param_shapes(cfg: VAEDecoderConfig) -> Dict[str, tuple]:
    out: Dict[str, tuple] = {}

    def conv(name, co, ci, k):
        out[name + ".weight"] = (co, ci, k, k)
        out[name + ".bias"] = (co,)

    def lin(name, co, ci):
        out[name + ".weight"] = (co, ci)
        out[name + ".bias"] = (co,)

    def norm(name, c):
        out[name + ".weight"] = (c,)
        out[name + ".bias"] = (c,)

    def resnet(p, ci, co):
        norm(p + ".norm1", ci)
        conv(p + ".conv1", co, ci, 3)
        norm(p + ".norm2", co)
        conv(p + ".conv2", co, co, 3)
        if ci != co:
            conv(p + ".conv_shortcut", co, ci, 1)

    top = cfg.chans[-1]
    conv("post_quant_conv", cfg.latent, cfg.latent, 1)
    conv("decoder.conv_in", top, cfg.latent, 3)
    resnet("decoder.mid_block.resnets.0", top, top)
    a = "decoder.mid_block.attentions.0"
    norm(a + ".group_norm", top)
    for nm in ("to_q", "to_k", "to_v", "to_out.0"):
        lin(f"{a}.{nm}", top, top)
    resnet("decoder.mid_block.resnets.1", top, top)
    rev = cfg.chans[::-1]
    prev = rev[0]
    for i, c in enumerate(rev):
        for j in range(cfg.layers + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", prev if j == 0 else c, c)
        if i != len(rev) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", c, c, 3)
        prev = c
    norm("decoder.conv_norm_out", cfg.chans[0])
    conv("decoder.conv_out", cfg.out_ch, cfg.chans[0], 3)
    return out


def _gn(sd, p, x, groups):
    return F.group_norm(x, groups, sd[p + ".weight"].to(x.dtype), sd[p + ".bias"].to(x.dtype), GN_EPS)


def _conv(sd, p, x, pad):
    return F.conv2d(x, sd[p + ".weight"].to(x.dtype), sd[p + ".bias"].to(x.dtype), padding=pad)


def resnet(sd: SD, p: str, x: torch.Tensor, groups: int) -> torch.Tensor:
    h = _conv(sd, p + ".conv1", F.silu(_gn(sd, p + ".norm1", x, groups)), 1)
    h = _conv(sd, p + ".conv2", F.silu(_gn(sd, p + ".norm2", h, groups)), 1)
    if p + ".conv_shortcut.weight" in sd:
        x = _conv(sd, p + ".conv_shortcut", x, 0)
    return x + h


def attention(sd: SD, p: str, x: torch.Tensor, groups: int) -> torch.Tensor:
    B, C, H, W = x.shape
    h = _gn(sd, p + ".group_norm", x.reshape(B, C, H * W), groups).transpose(1, 2)       # (B, L, C)

    def lin(nm, t):
        return F.linear(t, sd[f"{p}.{nm}.weight"].to(t.dtype), sd[f"{p}.{nm}.bias"].to(t.dtype))
    q, k, v = lin("to_q", h), lin("to_k", h), lin("to_v", h)
    prob = torch.softmax(q @ k.transpose(1, 2) * (C ** -0.5), dim=-1)                    # one head of C channels
    o = lin("to_out.0", prob @ v)
    return o.transpose(1, 2).reshape(B, C, H, W) + x


def decode(sd: SD, cfg: VAEDecoderConfig, z: torch.Tensor) -> torch.Tensor:
    """AutoencoderKL.decode(z).sample: z (B, latent, h, w) -> image (B, out_ch, 8h, 8w) for the 4-level default."""
    g = cfg.groups
    x = _conv(sd, "post_quant_conv", z, 0)
    x = _conv(sd, "decoder.conv_in", x, 1)
    x = resnet(sd, "decoder.mid_block.resnets.0", x, g)
    x = attention(sd, "decoder.mid_block.attentions.0", x, g)
    x = resnet(sd, "decoder.mid_block.resnets.1", x, g)
    n = len(cfg.chans)
    for i in range(n):
        for j in range(cfg.layers + 1):
            x = resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", x, g)
        if i != n - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = _conv(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", x, 1)
    x = F.silu(_gn(sd, "decoder.conv_norm_out", x, g))
    return _conv(sd, "decoder.conv_out", x, 1)


def to_uint8(image: torch.Tensor) -> torch.Tensor:
    """train_latent_diffusion.py:600-601, verbatim semantics."""
    image = (image / 2 + 0.5).clamp(0, 1)
    return (255 * image).to(torch.uint8)
