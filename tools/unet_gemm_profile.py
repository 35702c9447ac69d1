"""Per-shape time of every implicit-GEMM launch of one UNet forward (B=32, 64x64, bf16), from HIP events.
Usage: python tools/unet_gemm_profile.py [batch]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion_amd"))
sys.path.insert(0, ROOT)
from bench import init_weights_  # noqa: E402
from ldm import ops  # noqa: E402
from ldm.unet import UNet  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    dev = torch.device("cuda")
    with torch.device("meta"):
        net = UNet()
    net = net.to_empty(device=dev)
    init_weights_(net, 1)
    net.eval()
    x = torch.randn(B, 4, 64, 64, device=dev)
    ctx = torch.randn(B, 77, 768, device=dev).bfloat16()
    reps = 5
    with torch.inference_mode(), torch.autocast("cuda", dtype=torch.bfloat16):
        for _ in range(2):
            net(x, 500, ctx)
        log = []
        ops.set_launch_log(log)
        for _ in range(reps):
            net(x, 500, ctx)
        torch.cuda.synchronize()
        ops.set_launch_log(None)
    per = {}
    for name, flops, e0, e1, shape in log:
        d = per.setdefault((name, shape), [0, 0.0, 0.0])
        d[0] += 1
        d[1] += flops
        d[2] += e0.elapsed_time(e1) * 1e-3
    tot = sum(d[2] for d in per.values()) / reps
    print(f"GEMM time per forward {tot * 1e3:.2f} ms")
    print(f"{'tile':12s} {'M':>7s} {'N':>6s} {'K':>6s} k s u {'c1':>5s} epi  n/fwd  us/launch  TFLOP/s  ms/fwd")
    for (name, sh), (n, fl, t) in sorted(per.items(), key=lambda kv: -kv[1][2]):
        tile = name[name.index("<"):]
        print(f"{tile:12s} {sh[0]:7d} {sh[1]:6d} {sh[2]:6d} {sh[3]} {sh[4]} {sh[5]} {sh[6]:5d} {sh[7]:3d} {n // reps:5d} "
              f"{t / n * 1e6:9.1f} {fl / t / 1e12:8.1f} {t / reps * 1e3:7.3f}")


if __name__ == "__main__":
    main()
