"""Timing of the attention kernel on the UNet's shapes (B=32).  LDM_AMD_LIB=<other .so> runs another build (A/B).
Usage: python tools/attention_bench.py [--iters N] [--only "L0 self"] [--backward]
--backward times ldm_attention_backward (prep + dQ + dK/dV kernels; 10 B H Lq Lk d FLOP) instead of the forward and prints
its error against float32 autograd on the first sample."""
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
    if "--backward" in sys.argv:
        return backward(iters, only, dev, g)
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


def backward(iters, only, dev, g):
    import torch.nn.functional as F
    for name, B, lq, lk, H, d in CASES:
        if only is not None and only not in name:
            continue
        q = torch.randn(B, lq, H * d, device=dev, generator=g).bfloat16()
        kv = torch.randn(B, lk, 2 * H * d, device=dev, generator=g).bfloat16()
        do = torch.randn(B, lq, H * d, device=dev, generator=g).bfloat16()
        out, lse = ops.attention_train(q, kv, H)
        for _ in range(2):
            dq, dkv = ops.attention_backward(q, kv, out, do, lse, H)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            ops.attention_backward(q, kv, out, do, lse, H, dq=dq, dkv=dkv)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters

        def heads(t):
            return t.reshape(1, t.shape[1], H, d).transpose(1, 2)
        qr = q[:1].float().requires_grad_(True)
        kvr = kv[:1].float().requires_grad_(True)
        k_, v_ = kvr.chunk(2, dim=-1)
        o = F.scaled_dot_product_attention(heads(qr), heads(k_), heads(v_)).transpose(1, 2).reshape(1, lq, H * d)
        o.backward(do[:1].float())

        def rel(a, b):
            return ((a.float() - b).norm() / b.norm()).item()
        C = H * d
        print(f"{name:24s} {us:8.1f} us {10.0 * B * H * lq * lk * d / us / 1e6:7.1f} TFLOP/s   rel-L2 vs fp32 autograd: "
              f"dq {rel(dq[:1], qr.grad):.2e} dk {rel(dkv[:1, :, :C], kvr.grad[..., :C]):.2e} dv {rel(dkv[:1, :, C:], kvr.grad[..., C:]):.2e}",
              flush=True)


if __name__ == "__main__":
    main()
