"""Probe: the headline step (B = 32, 64x64, bf16 autocast) as eager launches, as a hipGraph replay, and as a hipGraph whose
forward runs as two concurrent half-batch branches (UNet.micro_batch_streams = 2 under capture).  Same-process A/B.
Usage: python tools/graph_streams_probe.py [steps]"""
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


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    dev = torch.device("cuda", 0)
    with torch.device("meta"):
        net = UNet()
    net = net.to_empty(device=dev).eval()
    init_weights_(net, 1234)
    sched = DDPMScheduler(device=dev)
    sched.set_timesteps(50, device=dev)
    g = torch.Generator(device=dev).manual_seed(100)
    lat = torch.randn(32, 4, 64, 64, device=dev, generator=g)
    ctx = torch.randn(32, 77, 768, device=dev, generator=g).bfloat16()
    noise = torch.randn(32, 4, 64, 64, device=dev, generator=g)
    ts = sched.timesteps

    def timeit(fn, tag):
        with torch.inference_mode(), torch.autocast("cuda", dtype=torch.bfloat16):
            for i in range(3):
                out = fn(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                out = fn(i)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / steps
        print(f"{tag:46s} {dt * 1e3:8.3f} ms/step  {32 / dt:8.1f} img-steps/s", flush=True)
        return out

    def eager(i):
        t = ts[i % len(ts)]
        return sched.step(net(lat, t, ctx), t, lat, noise=noise).prev_sample

    ref = timeit(eager, "eager, one stream")
    results = {}
    for nmb in (1, 2, 4):
        net.micro_batch_streams = nmb
        net._ldm_allow_capture_streams = True
        try:
            gr = GraphedUNet(net, 32, (64, 64))
        except Exception as exc:  # noqa: BLE001
            print(f"graph with {nmb} branch(es): capture failed: {type(exc).__name__}: {exc}")
            continue

        def graphed(i, gr=gr):
            t = ts[i % len(ts)]
            return sched.step(gr(lat, t, ctx), t, lat, noise=noise).prev_sample

        out = timeit(graphed, f"hipGraph replay, {nmb} branch(es)")
        results[nmb] = out
        print(f"   max |diff| vs eager: {float((out - ref).abs().max()):.3e}")
        del gr
    net.micro_batch_streams = 1
    timeit(eager, "eager, one stream (again)")


if __name__ == "__main__":
    main()
