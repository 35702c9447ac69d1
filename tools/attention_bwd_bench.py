#!/usr/bin/env python3
"""Attention backward micro-benchmark on the UNet's self-attention shapes at B=32 (prep + dQ + dK/dV kernels together)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion_amd"))
from ldm import ops  # noqa: E402

for name, B, H, L, Lk, d in [("L0 self d40", 32, 8, 4096, 4096, 40), ("L1 self d80", 32, 8, 1024, 1024, 80),
                             ("L2 self d160", 32, 8, 256, 256, 160), ("L0 cross d40", 32, 8, 4096, 77, 40),
                             ("L1 cross d80", 32, 8, 1024, 77, 80)]:
    C = H * d
    q = torch.randn(B, L, C, device="cuda").bfloat16()
    kv = torch.randn(B, Lk, 2 * C, device="cuda").bfloat16()
    do = torch.randn(B, L, C, device="cuda").bfloat16()
    out, lse = ops.attention_train(q, kv, H)
    for _ in range(2):
        ops.attention_backward(q, kv, out, do, lse, H)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.attention_backward(q, kv, out, do, lse, H)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 200
    fl = 10.0 * B * H * L * Lk * d
    print(f"{name:14s} {us:9.1f} us  {fl / us / 1e6:7.1f} TFLOP/s (10*h*L*Lk*d)", flush=True)
