"""Timing of the attention kernel on the UNet's shapes (B=32).  LDM_AMD_LIB=<other .so> runs another build (A/B).
Usage: python tools/attention_bench.py [--iters N] [--only "L0 self"]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion_amd"))
from ldm import ops  # noqa: E402

CASES = [("L0 self 4096x4096 d40", 32, 4096, 4096, 8, 40), ("L1 self 1024x1024 d80", 32, 1024, 1024, 8, 80),
         ("L2 self 256x256 d160", 32, 256, 256, 8, 160), ("L3 self 64x64 d160", 32, 64, 64, 8, 160),
         ("L0 cross 4096x77 d40", 32, 4096, 77, 8, 40), ("L1 cross 1024x77 d80", 32, 1024, 77, 8, 80),
         ("L2 cross 256x77 d160", 32, 256, 77, 8, 160)]


def main():
    iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 10
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None      # substring of a case name
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0)
    print("library:", os.environ.get("LDM_AMD_LIB", "default"))
    for name, B, lq, lk, H, d in CASES:
        if only is not None and only not in name:
            continue
        q = torch.randn(B, lq, H * d, device=dev, generator=g).bfloat16()
        kv = torch.randn(B, lk, 2 * H * d, device=dev, generator=g).bfloat16()
        for _ in range(3):
            ops.attention(q, kv, H)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.attention(q, kv, H)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
        print(f"{name:24s} {us:8.1f} us {4.0 * B * H * lq * lk * d / us / 1e6:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
