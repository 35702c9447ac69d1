#!/usr/bin/env python3
"""Training-step benchmark (BASELINE.json configs[3]): the reference's train step (train_latent_diffusion.py:1043-1126)
around the HIP UNet -- add_noise -> UNet forward (bf16 autocast, float32 master parameters) -> MSE -> backward (with the
bucketed gradient all-reduce of ldm/parallel.py overlapped when launched on N GPUs) -> clip_grad_norm_(1.0) -> AdamW.

    python tools/train_bench.py --batch 32 --steps 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        tools/train_bench.py --batch 32 --steps 5

UNet forward/backward and add_noise are the HIP kernels; the loss, the clip and the optimizer are the fused HIP tail of
ldm/optim.py or, with --torch-tail, the torch ops the reference's trainer calls.  Prints one JSON line: img/s over all ranks, ms per phase (HIP events on rank 0), the
fraction of the dense bf16 MFMA peak at 3x forward FLOPs per image, and the peak device memory."""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion_amd"))
sys.path.insert(0, ROOT)

from bench import FLOPS_PER_IMAGE_64, MFMA_BF16_DENSE_PEAK, init_weights_  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32, help="latents per GPU")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--bucket-mb", type=float, default=64.0)
    ap.add_argument("--small", action="store_true", help="config-1 network (plumbing check)")
    ap.add_argument("--torch-tail", action="store_true", help="loss / clip / AdamW with torch ops (the reference's own) instead "
                    "of the fused HIP tail (ldm/optim.py)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    from ldm.parallel import DataParallel
    from ldm.scheduling_ddpm import DDPMScheduler
    from ldm.unet import UNet

    kw = dict(block_out_channels=[32, 64, 64, 64]) if args.small else {}
    with torch.device("meta"):
        net = UNet(**kw)
    net = net.to_empty(device=dev)
    init_weights_(net, seed=1234)
    net.train()
    model = DataParallel(net, bucket_mb=args.bucket_mb)
    from ldm.optim import FusedAdamW, mse_loss
    if args.torch_tail:
        opt = torch.optim.AdamW(net.parameters(), lr=1e-5, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8, fused=True)
    else:
        opt = FusedAdamW(net.parameters(), lr=1e-5, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8, max_grad_norm=1.0)
    sched = DDPMScheduler(device=dev)

    B, L = args.batch, 64
    gen = torch.Generator(device=dev).manual_seed(100 + rank)
    lat = torch.randn(B, 4, L, L, device=dev, generator=gen)
    ctx = torch.randn(B, 77, 768, device=dev, generator=gen)

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(args.steps)]

    def step(i, rec=None):
        noise = torch.randn(lat.shape, device=dev, generator=gen)
        t = torch.randint(0, 1000, (B,), device=dev, generator=gen)
        if rec:
            rec[0].record()
        noisy = sched.add_noise(lat, noise, t)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            pred = model(noisy, t, ctx)
        if args.torch_tail:
            loss = F.mse_loss(pred.float(), noise.float(), reduction="mean")
            if rec:
                rec[1].record()
            loss.backward()
        else:
            loss, dpred = mse_loss(pred.detach(), noise)      # loss and its gradient in one pass
            if rec:
                rec[1].record()
            pred.backward(dpred)
        if rec:
            rec[2].record()
        if args.torch_tail:
            torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
        if rec:
            rec[3].record()
        opt.step()
        opt.zero_grad(set_to_none=True)
        if rec:
            rec[4].record()
        return loss

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for i in range(args.warmup):
        loss = step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i, ev[i])
    barrier()
    elapsed = time.perf_counter() - t0
    assert torch.isfinite(loss), "non-finite loss"
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax)
    if rank == 0:
        ph = [sum(e[k].elapsed_time(e[k + 1]) for e in ev) / args.steps for k in range(4)]
        ms = 1e3 * elapsed / args.steps
        flops = 3.0 * FLOPS_PER_IMAGE_64 * B
        print(json.dumps({
            "metric": "UNet training img/s at 64x64 latents (noise-prediction MSE, bf16 compute / fp32 master, AdamW)",
            "value": round(world * B * args.steps / elapsed, 2), "unit": "img/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms, 2), "per_gpu_batch": B, "dtype": "bf16", "data": "synthetic",
            "phases_ms": {"add_noise+forward+loss": round(ph[0], 2), "backward(+all-reduce)": round(ph[1], 2),
                          "clip_grad_norm": round(ph[2], 2), "adamw+zero_grad": round(ph[3], 2)},
            "frac_of_mfma_peak_at_3x_forward_flops": None if args.small else round(flops / (elapsed / args.steps) / MFMA_BF16_DENSE_PEAK, 4),
            "tail": "torch" if args.torch_tail else "fused (ldm/optim.py: clip folded into AdamW)",
            "collectives_per_backward": model.reducer.launched, "bucket_mb": args.bucket_mb,
            "peak_device_memory_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1),
            "loss": round(float(loss), 5)}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
