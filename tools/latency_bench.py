#!/usr/bin/env python3
"""Small-batch (serving) latency of one denoising step: eager launches against hipGraph replay (ldm/graph.py).
Default UNet, bf16, 4x64x64 latents; per batch size: ms per step (UNet forward + scheduler step), eager and graphed."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion_amd"))
sys.path.insert(0, ROOT)

from bench import init_weights_  # noqa: E402
from ldm.graph import GraphedUNet  # noqa: E402
from ldm.scheduling_ddpm import DDPMScheduler  # noqa: E402
from ldm.unet import UNet  # noqa: E402


def timed(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    dev = torch.device("cuda")
    with torch.device("meta"):
        net = UNet()
    net = net.to_empty(device=dev)
    init_weights_(net, seed=1234)
    net.eval()
    sch = DDPMScheduler(device=dev)
    sch.set_timesteps(50, device=dev)
    t = sch.timesteps[10]
    for B in (1, 2, 4, 8, 32):
        g = torch.Generator(device=dev).manual_seed(B)
        lat = torch.randn(B, 4, 64, 64, device=dev, generator=g)
        ctx = torch.randn(B, 77, 768, device=dev, generator=g).bfloat16()
        noise = torch.randn(B, 4, 64, 64, device=dev, generator=g)
        gnet = GraphedUNet(net, B, (64, 64))

        def eager():
            with torch.inference_mode(), torch.autocast("cuda", dtype=torch.bfloat16):
                return sch.step(net(lat, t, ctx), t, lat, noise=noise).prev_sample

        def graphed():
            with torch.inference_mode():
                return sch.step(gnet(lat, t, ctx), t, lat, noise=noise).prev_sample

        assert torch.equal(eager(), graphed())
        iters = 30 if B <= 8 else 10
        e, gr = timed(eager, iters), timed(graphed, iters)
        print(f"B={B:3d}  eager {e:7.2f} ms/step ({B / e * 1e3:7.1f} img-steps/s)   hipGraph {gr:7.2f} ms/step ({B / gr * 1e3:7.1f} img-steps/s)"
              f"   x{e / gr:.2f}", flush=True)


if __name__ == "__main__":
    main()
