"""Training-side input preparation around the hot path as fused kernels (train_latent_diffusion.py:1036-1079).

The reference's train step turns the VAE's (mean | logvar) moments into the UNet's inputs with a dozen small torch ops:
``DiagonalGaussianDistribution(moments).sample() * scaling_factor`` :1039-1040, ``randn_like`` + offset noise :1043-1048,
input perturbation :1049-1050, ``scheduler.add_noise`` :1062, and the text-conditioning dropout row select :1073-1079.
``prepare_inputs`` is one kernel for the first four, ``drop_conditioning`` one for the last.  The random draws are taken
with torch exactly where the reference takes them (same generators, same order), so a seeded run sees the same numbers.
``DiagonalGaussianDistribution`` lives in third-party diffusers (unpinned, absent offline): its published sampling formula
(logvar clamped to [-30, 20], std = exp(logvar / 2)) is restated -- "parity unpinned" for that line.
"""
from __future__ import annotations

from typing import NamedTuple, Optional

import torch

from . import _lib
from .ops import _gpu, _stream, dt_code


class TrainInputs(NamedTuple):
    latents: torch.Tensor          # scaling_factor * posterior sample, float32
    noise: torch.Tensor            # the loss target (with offset noise folded in), float32
    noisy_latents: torch.Tensor    # q(x_t | x_0), float32
    timesteps: torch.Tensor        # int64 (B,)


def prepare_inputs(scheduler, moments: torch.Tensor, scaling_factor: float, *, noise_offset: float = 0.0,
                   input_perturbation: float = 0.0, generator: Optional[torch.Generator] = None,
                   timesteps: Optional[torch.Tensor] = None) -> TrainInputs:
    """moments: (B, 2C, H, W) from ``vae._encode`` (mean | logvar).  Draw order as in the reference: posterior sample,
    noise, offset, perturbation, timesteps."""
    _gpu(moments)
    m = moments.to(torch.float32).contiguous()
    B, C2, H, W = m.shape
    C = C2 // 2
    dev = m.device
    shape = (B, C, H, W)
    eps = torch.randn(shape, device=dev, dtype=torch.float32, generator=generator)
    noise = torch.randn(shape, device=dev, dtype=torch.float32, generator=generator)
    offset = torch.randn((B, C, 1, 1), device=dev, generator=generator) if noise_offset else None
    extra = torch.randn(shape, device=dev, dtype=torch.float32, generator=generator) if input_perturbation else None
    if timesteps is None:
        timesteps = torch.randint(0, scheduler.config.num_train_timesteps, (B,), dtype=torch.int64, device=dev, generator=generator)
    t = timesteps.to(torch.long).contiguous()
    latents = torch.empty(shape, dtype=torch.float32, device=dev)
    noisy = torch.empty(shape, dtype=torch.float32, device=dev)
    tab = scheduler._tables_on(dev)
    with torch.cuda.device(dev):
        _lib.check(_lib.load().ldm_train_input_prep(
            m.data_ptr(), eps.data_ptr(), noise.data_ptr(), 0 if offset is None else offset.data_ptr(),
            0 if extra is None else extra.data_ptr(), t.data_ptr(), tab["sqrt_alphas_cumprod"].data_ptr(),
            tab["sqrt_one_minus_alphas_cumprod"].data_ptr(), scheduler.num_train_timesteps, B, C, H * W, float(scaling_factor),
            float(noise_offset), float(input_perturbation), latents.data_ptr(), noisy.data_ptr(), _stream(m)), "ldm_train_input_prep")
    return TrainInputs(latents, noise, noisy, t)


def drop_conditioning(enc_h: torch.Tensor, null_emb: torch.Tensor, drop_mask: torch.Tensor) -> torch.Tensor:
    """enc_h[drop_mask] = null_emb, in place (:1076-1079) without the host sync of ``drop_mask.any()``."""
    _gpu(enc_h, null_emb, drop_mask)
    if not enc_h.is_contiguous():
        raise ValueError("drop_conditioning: enc_h must be contiguous")
    B = enc_h.shape[0]
    per = enc_h.numel() // max(B, 1)
    ne = null_emb.to(enc_h.dtype).contiguous()
    if ne.numel() != per:
        raise ValueError("drop_conditioning: null embedding does not match one sample of enc_h")
    mask = drop_mask.to(torch.uint8).contiguous()
    with torch.cuda.device(enc_h.device):
        _lib.check(_lib.load().ldm_cond_dropout(enc_h.data_ptr(), ne.data_ptr(), mask.data_ptr(), dt_code(enc_h.dtype), B, per,
                                                _stream(enc_h)), "ldm_cond_dropout")
    return enc_h
