#!/usr/bin/env python3
"""Weight-gradient kernel micro-benchmark on the UNet's layer shapes at B=32 (LDM_WGRAD_V1=1: the 128x128 kernel)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion_amd"))
from ldm import ops  # noqa: E402

# (name, B, H, W, c0, c1, N, ksize, stride, upsample)
SHAPES = [
    ("L0 conv3x3 320->320", 32, 64, 64, 320, 0, 320, 3, 1, False),
    ("L1 conv3x3 640->640", 32, 32, 32, 640, 0, 640, 3, 1, False),
    ("L2 conv3x3 1280->1280", 32, 16, 16, 1280, 0, 1280, 3, 1, False),
    ("L3 conv3x3 1280->1280", 32, 8, 8, 1280, 0, 1280, 3, 1, False),
    ("L0 cat conv3x3 640->320", 32, 64, 64, 320, 320, 320, 3, 1, False),
    ("L2 cat conv3x3 2560->1280", 32, 16, 16, 1280, 1280, 1280, 3, 1, False),
    ("L0 ff in 320->2560", 32, 64, 64, 320, 0, 2560, 1, 1, False),
    ("L0 ff out 1280->320", 32, 64, 64, 1280, 0, 320, 1, 1, False),
    ("L0 qkv 320->960", 32, 64, 64, 320, 0, 960, 1, 1, False),
    ("L1 ff in 640->5120", 32, 32, 32, 640, 0, 5120, 1, 1, False),
    ("L0 proj 320->320", 32, 64, 64, 320, 0, 320, 1, 1, False),
    ("L1 down s2 640->640", 32, 32, 32, 640, 0, 640, 3, 2, False),
    ("L1 up 1280->1280 @32", 32, 16, 16, 1280, 0, 1280, 3, 1, True),
]


def main():
    dev = torch.device("cuda")
    tot = 0.0
    for name, B, H, W, c0, c1, N, ks, stride, up in SHAPES:
        a0 = torch.randn(B, H, W, c0, device=dev).bfloat16()
        a1 = torch.randn(B, H, W, c1, device=dev).bfloat16() if c1 else None
        hs, ws = (2 * H, 2 * W) if up else (H, W)
        ho, wo = (hs + 2 * (ks // 2) - ks) // stride + 1, (ws + 2 * (ks // 2) - ks) // stride + 1
        dy = torch.randn(B, ho, wo, N, device=dev).bfloat16()
        if ks == 1:
            a0v, dyv = a0.view(-1, c0), dy.view(-1, N)
        else:
            a0v, dyv = a0, dy
        for _ in range(2):
            ops.conv_wgrad(a0v, dyv, a1=a1, ksize=ks, stride=stride, upsample=up)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.conv_wgrad(a0v, dyv, a1=a1, ksize=ks, stride=stride, upsample=up)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 5
        fl = 2.0 * B * ho * wo * N * ks * ks * (c0 + c1)
        tot += us
        print(f"{name:28s} M={B*ho*wo:7d} N={N:5d} K={ks*ks*(c0+c1):6d}  {us:8.1f} us  {fl/us/1e6:7.1f} TFLOP/s", flush=True)
    print(f"sum {tot:.1f} us")


if __name__ == "__main__":
    main()
