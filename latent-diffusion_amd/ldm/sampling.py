"""Latent sampling loop around the hot path, sharded over GPUs by latent.

Counterpart of the reverse-diffusion loop of ``generate`` (ldm/tools/train_latent_diffusion.py:563-593) without the
tokenizer / text encoder around it (out of scope: SURVEY.md section 8): ``set_timesteps`` -> for every timestep one
denoiser forward (two batched forwards' worth with classifier-free guidance, :581-592) -> ``scheduler.step``; then
(``decode_images``, :596-601) the VAE decoder of ldm/vae.py and the uint8 conversion.

Multi-GPU: every latent is independent through all steps, so rank g of G owns the contiguous slice
``shard_range(N, g, G)`` of the batch and the loop contains NO collective.  So that the result does not depend on G,
the initial latents and the per-step noise are functions of (seed, step, global sample index, element): a counter-based
generator (``ldm_gaussian_noise``: Philox4x32-10 keyed by the seed, counter = element block / sample / step) from which a
rank draws ONLY its own rows -- nothing is drawn for the other ranks' samples.  ``gather_latents`` is the optional final
all-gather.
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


def global_noise(seed: int, step: int, shape, lo: int, hi: int, device, dtype=torch.float32) -> torch.Tensor:
    """Rows [lo, hi) of the N(0,1) tensor of `shape` (shape[0] = global batch) that belongs to (seed, step); only those rows
    are drawn (ops.gaussian_noise).  step = -1: the initial latents."""
    from . import ops
    z = ops.gaussian_noise(seed, step, lo, hi - lo, tuple(shape[1:]), device)
    return z if dtype == torch.float32 else z.to(dtype)


@torch.inference_mode()
def sample_latents(denoiser: Callable, scheduler, ctx: torch.Tensor, *, latent_shape=(4, 64, 64),
                   num_inference_steps: int = 50, guidance_scale: Optional[float] = None,
                   uncond_ctx: Optional[torch.Tensor] = None, seed: int = 0, rank: int = 0, world: int = 1,
                   device=None, latent_dtype: torch.dtype = torch.float32, micro_batch: Optional[int] = None,
                   noise_fn: Optional[Callable] = None) -> torch.Tensor:
    """Run the reverse loop for this rank's shard of the batch and return its final latents (hi-lo, C, H, W).

    ctx: (N, S, D) conditioning of the WHOLE batch (each rank slices its rows).  With ``guidance_scale`` the
    denoiser sees [latents, latents] with [uncond_ctx, ctx] exactly as the reference does (:581, :561) and the
    combine eps_u + g (eps_c - eps_u) (:592) runs inside the scheduler's fused step kernel.
    micro_batch: run the shard as independent chunks of at most this many latents (each through all steps); the result
    is the same, the activations stay small (a 256-latent shard in one batch would pass the kernels' 2 GiB operands).
    noise_fn(seed, step, shape, lo, hi, device, dtype): the noise source, default ``global_noise`` (the HIP generator); the
    host-logic tests, which run this loop on CPU tensors with a stand-in denoiser, pass the oracle's restatement of it.
    """
    noise_fn = noise_fn or global_noise
    n = ctx.shape[0]
    lo, hi = shard_range(n, rank, world)
    if micro_batch is not None and hi - lo > micro_batch:
        parts = []
        for c0 in range(lo, hi, micro_batch):
            parts.append(_sample_rows(denoiser, scheduler, ctx, c0, min(c0 + micro_batch, hi), latent_shape, num_inference_steps,
                                      guidance_scale, uncond_ctx, seed, device, latent_dtype, noise_fn))
        return torch.cat(parts, dim=0)
    return _sample_rows(denoiser, scheduler, ctx, lo, hi, latent_shape, num_inference_steps, guidance_scale, uncond_ctx, seed,
                        device, latent_dtype, noise_fn)


def _sample_rows(denoiser, scheduler, ctx, lo, hi, latent_shape, num_inference_steps, guidance_scale, uncond_ctx, seed, device,
                 latent_dtype, noise_fn):
    """The reverse loop for global rows [lo, hi)."""
    n = ctx.shape[0]
    device = device if device is not None else ctx.device
    my_ctx = ctx[lo:hi].to(device)
    scheduler.set_timesteps(num_inference_steps, device=device)
    shape = (n, *latent_shape)
    latents = noise_fn(seed, -1, shape, lo, hi, device, latent_dtype) * scheduler.init_noise_sigma
    if guidance_scale is not None:
        if uncond_ctx is None:
            raise ValueError("classifier-free guidance needs uncond_ctx ((1, S, D) or (N, S, D))")
        unc = uncond_ctx.to(device)
        unc = unc.expand(n, -1, -1)[lo:hi] if unc.shape[0] == 1 else unc[lo:hi]
        both_ctx = torch.cat([unc, my_ctx], dim=0)
    for i, t in enumerate(scheduler.timesteps):
        noise = noise_fn(seed, i, shape, lo, hi, device, torch.float32)
        if guidance_scale is None:
            eps = denoiser(scheduler.scale_model_input(latents, t), t, my_ctx)
            latents = scheduler.step(eps, t, latents, noise=noise).prev_sample
        else:
            latent_in = scheduler.scale_model_input(torch.cat([latents, latents], dim=0), t)
            eps_u, eps_c = denoiser(latent_in, t, both_ctx).chunk(2)
            latents = scheduler.step(eps_u, t, latents, noise=noise, guidance=(eps_c, guidance_scale)).prev_sample
    return latents


@torch.inference_mode()
def decode_images(vae, latents: torch.Tensor, *, compute_dtype: Optional[torch.dtype] = torch.bfloat16,
                  micro_batch: Optional[int] = None) -> torch.Tensor:
    """train_latent_diffusion.py:596-601 for this rank's latents: ``latents / scaling_factor`` -> ``vae.decode`` ->
    ``(255 * (x / 2 + 0.5).clamp(0, 1)).to(uint8)``.  Returns (n, 3, 8h, 8w) uint8.  ``vae``: ldm.vae.AutoencoderKL.
    compute_dtype bfloat16 = what ``accelerator.autocast()`` around ``generate`` gives the reference (:626); None = the
    latents' own dtype."""
    from .vae import to_uint8
    z = latents / vae.config.scaling_factor
    if compute_dtype == torch.bfloat16 and z.dtype != torch.bfloat16:
        with torch.autocast("cuda", dtype=torch.bfloat16):
            img = vae.decode(z, return_dict=False, micro_batch=micro_batch)[0]
    else:
        img = vae.decode(z, return_dict=False, micro_batch=micro_batch)[0]
    return to_uint8(img)


def generate_images(denoiser: Callable, scheduler, vae, ctx: torch.Tensor, **kw) -> torch.Tensor:
    """``generate`` (train_latent_diffusion.py:529-607) from the text embeddings on: the reverse loop for this rank's
    shard, then the decoder and the uint8 conversion.  Keyword arguments are ``sample_latents``'."""
    return decode_images(vae, sample_latents(denoiser, scheduler, ctx, **kw))


def gather_latents(local: torch.Tensor, n: int, rank: int, world: int, group=None, force: bool = False) -> torch.Tensor:
    """All-gather the shards back into the (n, ...) batch (the only collective of the sampling path, after the loop).
    force: run the collective also in a group of one rank (tests: drives RCCL on a one-GPU box)."""
    if world == 1 and not force:
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
