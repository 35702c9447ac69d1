#!/usr/bin/env python3
"""Headline benchmark: UNet denoise img-steps/sec at 64x64 latents on MI355X (BASELINE.json metric).

One "step" = one pass of the hot path over one batch: ``UNet.forward`` (SD-1.5-shaped, bf16 storage / fp32
accumulate) on 32 latents of 4x64x64 with a 77x768 text context, followed by ``DDPMScheduler.step`` on the
result -- BASELINE.json configs[1], the configuration the metric is quoted on.  Inputs are resident in HBM before
the timed region.  With --gpus N every rank runs the same per-GPU workload on its own batch shard (independent
latents, no data-path collective: weak scaling); value = images x steps over all ranks / max-over-ranks time.

Printed JSON (rank 0, one line) also carries
  roofline      the dominant kernel (bf16 implicit-GEMM conv/linear): algorithmic FLOPs per launch / its average
                launch duration, measured with HIP events on the launch stream inside the timed region,
                against the dense bf16 MFMA peak;
  cpu_baseline  the CPU oracle (oracle/unet_ref.py + scheduler_ref.py: a port, not the product) timed on this
                host's cores on a bounded sample of the same workload (N=1 only).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion_amd"))
sys.path.insert(0, ROOT)

MFMA_BF16_DENSE_PEAK = 2.5e15       # FLOP/s, /opt/skills/guides/MI355X_MICROARCH.md (dense; 2:1-sparse figure is 5e15)
FLOPS_PER_IMAGE_64 = 803273441280.0  # algorithmic forward FLOPs per latent at 64x64 (tests/golden/meta.json)
BATCH, LATENT, CTX_LEN, CTX_DIM = 32, 64, 77, 768


def init_weights_(net: torch.nn.Module, seed: int):
    """Random-init weights of the real architecture, drawn on the device (no checkpoints offline)."""
    g = torch.Generator(device="cuda").manual_seed(seed)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if p.dim() >= 2:
                fan_in = p[0].numel()
                p.normal_(0.0, 1.0 / math.sqrt(fan_in), generator=g)
            elif name.endswith("weight"):
                p.fill_(1.0)
            else:
                p.normal_(0.0, 0.02, generator=g)


def cpu_baseline(sample_forwards: int = 12):
    """Oracle on the host cores: default UNet fp32 forward at B=1, 64x64 + one scheduler step (bounded sample)."""
    from oracle import scheduler_ref as sref
    from oracle import unet_ref as uref
    cfg = uref.UNetConfig()
    shapes = uref.param_shapes(cfg)
    g = torch.Generator().manual_seed(0)
    sd = {}
    for k, shp in shapes.items():  # cheap fill (values do not matter for timing)
        sd[k] = torch.empty(shp).normal_(0, 0.02, generator=g) if len(shp) > 1 else torch.ones(shp)
    x = torch.randn(1, 4, LATENT, LATENT, generator=g)
    ctx = torch.randn(1, CTX_LEN, CTX_DIM, generator=g)
    tabs = sref.build_tables()
    noise = torch.randn(1, 4, LATENT, LATENT, generator=g)
    with torch.no_grad():
        uref.unet_forward(sd, cfg, x, 500, ctx)  # warm-up
        t0 = time.perf_counter()
        for _ in range(sample_forwards):
            eps = uref.unet_forward(sd, cfg, x, 500, ctx)
            sref.step(tabs, eps, 500, 480, x, noise)
        dt = time.perf_counter() - t0
    return {"value": sample_forwards / dt, "unit": "img-steps/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{sample_forwards} forwards of the default UNet (fp32, B=1, 4x64x64, ctx 77x768) + scheduler step "
                      f"after 1 warm-up, {dt:.1f} s; os.cpu_count()={os.cpu_count()}"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH, help="latents per GPU (default: the metric's 32)")
    ap.add_argument("--latent", type=int, default=LATENT, help="latent extent (default 64: the metric's; 96 = BASELINE configs[4])")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE {world}")
    if os.environ.get("LDM_BENCH_SHARE_GPU") == "1":   # rehearsal only: every rank on device 0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # RCCL ("nccl" on ROCm) is the backend of record; LDM_BENCH_BACKEND=gloo with LDM_BENCH_SHARE_GPU=1 only
        # rehearses the launch contract with several ranks on a one-GPU box (RCCL refuses two ranks on one device).
        backend = os.environ.get("LDM_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from ldm import ops
    from ldm.scheduling_ddpm import DDPMScheduler
    from ldm.unet import UNet

    with torch.device("meta"):
        net = UNet()
    net = net.to_empty(device=dev)
    init_weights_(net, seed=1234)
    net.eval()
    sched = DDPMScheduler(device=dev)
    sched.set_timesteps(50, device=dev)

    B = args.batch
    LAT = args.latent
    if LAT == 64:
        flops_per_image = FLOPS_PER_IMAGE_64
    else:   # algorithmic FLOPs of another extent: the oracle's shape walk (host arithmetic only, nothing is executed)
        from oracle import unet_ref as uref
        flops_per_image = uref.flops_per_image(uref.UNetConfig(), LAT, LAT)
    gen = torch.Generator(device=dev).manual_seed(100 + rank)  # every rank owns different latents (batch shard)
    lat0 = torch.randn(B, 4, LAT, LAT, device=dev, generator=gen)
    ctx = torch.randn(B, CTX_LEN, CTX_DIM, device=dev, generator=gen).bfloat16()
    noise = torch.randn(B, 4, LAT, LAT, device=dev, generator=gen)
    ts = sched.timesteps

    def one_step(lat, i):
        t = ts[i % len(ts)]
        eps = net(lat, t, ctx)                     # bf16 UNet forward (autocast; conv_in reads the fp32 latents directly)
        return sched.step(eps, t, lat, noise=noise).prev_sample  # fp32 scheduler update, fused kernel

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with torch.inference_mode(), torch.autocast("cuda", dtype=torch.bfloat16):
        lat = lat0
        for i in range(args.warmup):
            lat = one_step(lat0, i)
        log = []
        ops.set_launch_log(log)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            lat = one_step(lat0, i)
        barrier()
        elapsed = time.perf_counter() - t0
        ops.set_launch_log(None)
    assert torch.isfinite(lat).all(), "non-finite latents"

    if dist is not None:
        tmax = torch.tensor([elapsed], device=dev if dist.get_backend() == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax)

    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        value = world * B * args.steps / elapsed
        # dominant kernel from the live event log
        per = {}
        for name, flops, e0, e1, _shape in log:
            d = per.setdefault(name, [0, 0.0, 0.0])
            d[0] += 1
            d[1] += flops
            d[2] += e0.elapsed_time(e1) * 1e-3
        gemm_time = sum(d[2] for d in per.values())
        dom = max(per, key=lambda k: per[k][2])
        n_l, fl, tt = per[dom]
        achieved = fl / tt
        # HBM bytes per launch of the dominant kernel: not measurable from inside the process; the figure is the one the
        # separate rocprofv3 --pmc passes over this same command gave (FETCH_SIZE, doubled per the gfx950 correction, +
        # WRITE_SIZE; profiles/r1_v10_pmc_per_kernel.md), reported only for the shape it was measured on.
        pmc_traffic = {"conv_gemm_bf16_kernel<256,160,64>": 288.2e6}
        traffic = pmc_traffic.get(dom) if (LAT == 64 and B == BATCH) else None
        out = {
            "metric": f"UNet denoise img-steps/sec at {LAT}x{LAT} latents",
            "value": round(value, 2), "unit": "img-steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "SD-1.5-shaped UNet (859.5M params, random init) bf16 forward + DDPM scheduler step, "
                                   f"batch {B} of 4x{LAT}x{LAT} latents + 77x768 context per GPU "
                                   + ("(BASELINE.json configs[1])" if LAT == 64 and B == BATCH else "(non-headline shape)"),
                       "per_gpu_batch": B, "global_batch": B * world, "latent": f"4x{LAT}x{LAT}",
                       "parallelism": f"batch-sharded x{world}, no data-path collective",
                       "end_to_end_tflops_per_gpu": round(flops_per_image * B / (elapsed / args.steps) / 1e12, 1),
                       "end_to_end_frac_of_mfma_peak": round(flops_per_image * B / (elapsed / args.steps) / MFMA_BF16_DENSE_PEAK, 4),
                       "gemm_kernel_share_of_step": round(gemm_time / elapsed, 3)},
            "roofline": {"bound": "mfma", "kernel": dom, "achieved": round(achieved / 1e12, 1), "peak": MFMA_BF16_DENSE_PEAK / 1e12,
                         "unit": "TFLOP/s", "frac": round(achieved / MFMA_BF16_DENSE_PEAK, 4), "traffic": traffic,
                         "traffic_note": "HBM+Infinity-Cache bytes per launch from committed PMC passes "
                                         "(profiles/r1_v10_pmc_per_kernel.md), not collected live",
                         "launches": n_l, "avg_launch_us": round(1e6 * tt / n_l, 1),
                         "algorithmic_gflop_per_launch": round(fl / n_l / 1e9, 2),
                         "all_gemm_kernels": {k: {"launches": v[0], "tflops": round(v[1] / v[2] / 1e12, 1),
                                                  "avg_us": round(1e6 * v[2] / v[0], 1)} for k, v in sorted(per.items())}},
        }
        if world == 1 and not args.no_cpu_baseline and LAT == 64:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
