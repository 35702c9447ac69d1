"""Latent sampling loop around the hot path, sharded over GPUs by latent.

Counterpart of the reverse-diffusion loop of ``generate`` (ldm/tools/train_latent_diffusion.py:563-593) without the
tokenizer / text encoder / VAE around it (out of scope: SURVEY.md section 8): ``set_timesteps`` -> for every timestep
one denoiser forward (two batched forwards' worth with classifier-free guidance, :581-592) -> ``scheduler.step``.

Multi-GPU: every latent is independent through all steps, so rank g of G owns the contiguous slice
``shard_range(N, g, G)`` of the batch and the loop contains NO collective.  So that the result does not depend on G,
the initial latents and the per-step noise are functions of (seed, step, global sample index): each rank draws the
step's noise for the whole batch from a generator seeded by (seed, step) and keeps its rows (4 M floats at N = 256,
negligible beside a forward).  ``gather_latents`` is the optional final all-gather.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous rows [lo, hi) of an n-row batch owned by `rank`; the first n % world ranks take one extra row."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def _seed_for(seed: int, step: int) -> int:
    # distinct, reproducible 63-bit seeds per (seed, step); step -1 = initial latents
    return (seed * 1_000_003 + (step + 1) * 7_919 + 0x5DEECE66D) & 0x7FFF_FFFF_FFFF_FFFF


def global_noise(seed: int, step: int, shape, lo: int, hi: int, device, dtype=torch.float32) -> torch.Tensor:
    """Rows [lo, hi) of the N(0,1) tensor of `shape` (shape[0] = global batch) that belongs to (seed, step).
    Drawn on `device` for the whole batch, so the values of a row do not depend on how the batch is sharded."""
    g = torch.Generator(device=device).manual_seed(_seed_for(seed, step))
    full = torch.randn(tuple(shape), generator=g, device=device, dtype=dtype)
    return full[lo:hi].contiguous()


@torch.inference_mode()
def sample_latents(denoiser: Callable, scheduler, ctx: torch.Tensor, *, latent_shape=(4, 64, 64),
                   num_inference_steps: int = 50, guidance_scale: Optional[float] = None,
                   uncond_ctx: Optional[torch.Tensor] = None, seed: int = 0, rank: int = 0, world: int = 1,
                   device=None, latent_dtype: torch.dtype = torch.float32, micro_batch: Optional[int] = None) -> torch.Tensor:
    """Run the reverse loop for this rank's shard of the batch and return its final latents (hi-lo, C, H, W).

    ctx: (N, S, D) conditioning of the WHOLE batch (each rank slices its rows).  With ``guidance_scale`` the
    denoiser sees [latents, latents] with [uncond_ctx, ctx] exactly as the reference does (:581, :561) and the
    combine eps_u + g (eps_c - eps_u) (:592) runs inside the scheduler's fused step kernel.
    micro_batch: run the shard as independent chunks of at most this many latents (each through all steps); the result
    is the same, the activations stay small (a 256-latent shard in one batch would pass the kernels' 2 GiB operands).
    """
    n = ctx.shape[0]
    lo, hi = shard_range(n, rank, world)
    if micro_batch is not None and hi - lo > micro_batch:
        parts = []
        for c0 in range(lo, hi, micro_batch):
            parts.append(_sample_rows(denoiser, scheduler, ctx, c0, min(c0 + micro_batch, hi), latent_shape, num_inference_steps,
                                      guidance_scale, uncond_ctx, seed, device, latent_dtype))
        return torch.cat(parts, dim=0)
    return _sample_rows(denoiser, scheduler, ctx, lo, hi, latent_shape, num_inference_steps, guidance_scale, uncond_ctx, seed,
                        device, latent_dtype)


def _sample_rows(denoiser, scheduler, ctx, lo, hi, latent_shape, num_inference_steps, guidance_scale, uncond_ctx, seed, device,
                 latent_dtype):
    """The reverse loop for global rows [lo, hi)."""
    n = ctx.shape[0]
    device = device if device is not None else ctx.device
    my_ctx = ctx[lo:hi].to(device)
    scheduler.set_timesteps(num_inference_steps, device=device)
    shape = (n, *latent_shape)
    latents = global_noise(seed, -1, shape, lo, hi, device, latent_dtype) * scheduler.init_noise_sigma
    if guidance_scale is not None:
        if uncond_ctx is None:
            raise ValueError("classifier-free guidance needs uncond_ctx ((1, S, D) or (N, S, D))")
        unc = uncond_ctx.to(device)
        unc = unc.expand(n, -1, -1)[lo:hi] if unc.shape[0] == 1 else unc[lo:hi]
        both_ctx = torch.cat([unc, my_ctx], dim=0)
    for i, t in enumerate(scheduler.timesteps):
        noise = global_noise(seed, i, shape, lo, hi, device, torch.float32)
        if guidance_scale is None:
            eps = denoiser(scheduler.scale_model_input(latents, t), t, my_ctx)
            latents = scheduler.step(eps, t, latents, noise=noise).prev_sample
        else:
            latent_in = scheduler.scale_model_input(torch.cat([latents, latents], dim=0), t)
            eps_u, eps_c = denoiser(latent_in, t, both_ctx).chunk(2)
            latents = scheduler.step(eps_u, t, latents, noise=noise, guidance=(eps_c, guidance_scale)).prev_sample
    return latents


def gather_latents(local: torch.Tensor, n: int, rank: int, world: int, group=None) -> torch.Tensor:
    """All-gather the shards back into the (n, ...) batch (the only collective of the sampling path, after the loop)."""
    if world == 1:
        return local
    import torch.distributed as dist
    sizes = [shard_range(n, r, world) for r in range(world)]
    pad = max(hi - lo for lo, hi in sizes)
    buf = local.new_zeros((pad, *local.shape[1:]))
    buf[: local.shape[0]] = local
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    return torch.cat([p[: hi - lo] for p, (lo, hi) in zip(parts, sizes)], dim=0)


def max_over_ranks(seconds: float, device, world: int, group=None) -> float:
    """Slowest rank's wall time: what bench.py divides the job's img-steps by."""
    if world == 1:
        return seconds
    import torch.distributed as dist
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t)
