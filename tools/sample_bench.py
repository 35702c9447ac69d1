#!/usr/bin/env python3
"""BASELINE.json configs[2]: batch-sharded DDPM sampling -- 256 latents of 4x64x64, 50 steps, default UNet in bf16, the
batch split over the ranks (ldm/sampling.py: no collective inside the loop, noise a function of the global sample index,
one all-gather at the end).

    python tools/sample_bench.py                      # 1 GPU: 12,800 img-steps
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P tools/sample_bench.py

Prints one JSON line from rank 0: img-steps/s over the whole job (slowest rank's wall time), plus a checksum of the
gathered latents that must not depend on N."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion_amd"))
sys.path.insert(0, ROOT)

from bench import init_weights_  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--latents", type=int, default=256)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--micro-batch", type=int, default=32)
    ap.add_argument("--guidance", type=float, default=None, help="classifier-free guidance scale (2 forwards' worth per step)")
    ap.add_argument("--decode", action="store_true", help="also run the VAE decoder + uint8 conversion on the final latents "
                    "(train_latent_diffusion.py:596-601) and report the whole generate() job in images/s")
    args = ap.parse_args()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)
    from ldm import sampling
    from ldm.scheduling_ddpm import DDPMScheduler
    from ldm.unet import UNet

    with torch.device("meta"):
        net = UNet()
    net = net.to_empty(device=dev)
    init_weights_(net, seed=1234)
    net.eval()
    g = torch.Generator(device=dev).manual_seed(7)
    ctx = torch.randn(args.latents, 77, 768, device=dev, generator=g).bfloat16()     # same on every rank
    unc = torch.randn(1, 77, 768, device=dev, generator=g).bfloat16()
    kw = dict(latent_shape=(4, 64, 64), num_inference_steps=args.steps, seed=11, rank=rank, world=world,
              micro_batch=args.micro_batch, guidance_scale=args.guidance, uncond_ctx=unc if args.guidance else None)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        sampling.sample_latents(net, DDPMScheduler(device=dev), ctx[: world * 2], **{**kw, "num_inference_steps": 2})  # warm-up
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        local = sampling.sample_latents(net, DDPMScheduler(device=dev), ctx, **kw)
        torch.cuda.synchronize(dev)
        elapsed = time.perf_counter() - t0
    extra = {}
    if args.decode:
        from ldm.vae import AutoencoderKL
        with torch.device("meta"):
            vae = AutoencoderKL()
        vae = vae.to_empty(device=dev)
        init_weights_(vae, seed=77)
        vae.eval()
        z = local.clamp(-4, 4)                  # random weights do not denoise: keep the decoder's input in a latent's range
        sampling.decode_images(vae, z[:2])      # warm-up (packs the filters)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        images = sampling.decode_images(vae, z, micro_batch=8)
        torch.cuda.synchronize(dev)
        dec = sampling.max_over_ranks(time.perf_counter() - t1, dev, world)
        assert images.dtype == torch.uint8 and images.shape[1:] == (3, 512, 512)
        extra = {"decode_seconds": round(dec, 3), "decode_images_per_s": round(args.latents / dec, 1),
                 "generate_images_per_s": round(args.latents / (sampling.max_over_ranks(elapsed, dev, world) + dec), 2)}
    elapsed = sampling.max_over_ranks(elapsed, dev, world)
    full = sampling.gather_latents(local, args.latents, rank, world)
    if rank == 0:
        print(json.dumps({"metric": "batch-sharded DDPM sampling img-steps/sec at 64x64 latents",
                          "value": round(args.latents * args.steps / elapsed, 2), "unit": "img-steps/s", "n_gpus": world,
                          "latents": args.latents, "steps": args.steps, "seconds": round(elapsed, 3), "dtype": "bf16",
                          "guidance": args.guidance, "micro_batch": args.micro_batch, "scaling": "strong",
                          "finite": bool(torch.isfinite(full).all()), "latent_abs_mean": round(float(full.abs().mean()), 6), **extra}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
