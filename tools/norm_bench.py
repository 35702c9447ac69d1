"""Achieved HBM rate of the GroupNorm / LayerNorm kernels on the UNet's shapes (B=32, bf16).
LDM_AMD_LIB=<other .so> runs another build (A/B).  Usage: python tools/norm_bench.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion_amd"))
from ldm import ops  # noqa: E402

B = 32


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def main():
    dev = torch.device("cuda")
    print("library:", os.environ.get("LDM_AMD_LIB", "default"))
    for H, C, C1 in [(64, 320, 0), (64, 640, 320), (32, 640, 0), (32, 1280, 640), (16, 1280, 0), (16, 1280, 1280), (8, 1280, 1280)]:
        x = torch.randn(B, H, H, C, device=dev).bfloat16()
        s = torch.randn(B, H, H, C1, device=dev).bfloat16() if C1 else None
        g = torch.ones(C + C1, device=dev)
        b = torch.zeros(C + C1, device=dev)
        us = timeit(lambda: ops.groupnorm(x, g, b, 32, a1=s, silu=True))
        nbytes = x.numel() * 2 * 3 + (s.numel() * 2 * 3 if C1 else 0)   # stats read + apply read + write
        print(f"groupnorm+silu {B}x{H}x{H}x({C}+{C1})  {us:7.1f} us  {nbytes / us / 1e3:7.1f} GB/s (3 passes of 2 B/elem)", flush=True)
    for L, C in [(4096, 320), (1024, 640), (256, 1280), (64, 1280)]:
        x = torch.randn(B, L, C, device=dev).bfloat16()
        g = torch.ones(C, device=dev)
        b = torch.zeros(C, device=dev)
        us = timeit(lambda: ops.layernorm(x, g, b))
        print(f"layernorm {B}x{L}x{C}  {us:7.1f} us  {x.numel() * 4 / us / 1e3:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
