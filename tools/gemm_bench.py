"""Micro-benchmark of the implicit-GEMM kernel on the UNet's characteristic shapes (B=32, 64x64 latents).
Usage: python tools/gemm_bench.py [case-substring] [--iters N]"""
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion_amd"))
from ldm import ops  # noqa: E402

B = 32
CASES = [  # name, (H, W, C0, C1), N, ksize, stride, up, glu
    ("L0 conv3x3 320->320", (64, 64, 320, 0), 320, 3, 1, False, None),
    ("L0 conv3x3 960->320 cat", (64, 64, 640, 320), 320, 3, 1, False, None),
    ("L1 conv3x3 640->640", (32, 32, 640, 0), 640, 3, 1, False, None),
    ("L2 conv3x3 1280->1280", (16, 16, 1280, 0), 1280, 3, 1, False, None),
    ("L3 conv3x3 1280->1280", (8, 8, 1280, 0), 1280, 3, 1, False, None),
    ("L3 conv3x3 2560->1280 cat", (8, 8, 1280, 1280), 1280, 3, 1, False, None),
    ("L0 down s2 320->320", (64, 64, 320, 0), 320, 3, 2, False, None),
    ("L0 up conv 640->640", (32, 32, 640, 0), 640, 3, 1, True, None),
    ("L0 qkv 320->960", (64, 64, 320, 0), 960, 1, 1, False, None),
    ("L0 geglu 320->2560", (64, 64, 320, 0), 2560, 1, 1, False, "geglu"),
    ("L0 ffout 1280->320", (64, 64, 1280, 0), 320, 1, 1, False, None),
    ("L0 proj 320->320", (64, 64, 320, 0), 320, 1, 1, False, None),
    ("L1 geglu 640->5120", (32, 32, 640, 0), 5120, 1, 1, False, "geglu"),
    ("L2 geglu 1280->10240", (16, 16, 1280, 0), 10240, 1, 1, False, "geglu"),
    # the same GEMMs with the cheaper gate (exp + rcp + 2 instead of the erf's ~17 + two quarter-rate): what the epilogue's
    # vector arithmetic costs the GLU launches
    ("L0 swiglu 320->2560", (64, 64, 320, 0), 2560, 1, 1, False, "swiglu"),
    ("L1 swiglu 640->5120", (32, 32, 640, 0), 5120, 1, 1, False, "swiglu"),
    ("L2 swiglu 1280->10240", (16, 16, 1280, 0), 10240, 1, 1, False, "swiglu"),
    ("ctx kv 768->640", (77, 1, 768, 0), 640, 1, 1, False, None),
    # the GLU projections' shapes WITHOUT the gate epilogue: what the tile alone is worth on them
    ("L0 ff1 plain 320->2560", (64, 64, 320, 0), 2560, 1, 1, False, None),
    ("L1 ff1 plain 640->5120", (32, 32, 640, 0), 5120, 1, 1, False, None),
    ("L2 ff1 plain 1280->10240", (16, 16, 1280, 0), 10240, 1, 1, False, None),
]


def main():
    skip = {i + 1 for i, a in enumerate(sys.argv) if a in ("--iters", "--tile")}
    sel = [a for i, a in enumerate(sys.argv) if i >= 1 and not a.startswith("--") and i not in skip]
    iters = 20
    if "--iters" in sys.argv:
        iters = int(sys.argv[sys.argv.index("--iters") + 1])
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0)
    tiles = [None]
    if "--sweep" in sys.argv:   # every BK=64 tile the library instantiates, forced through LDM_GEMM_TILE
        tiles = [None, "64,128", "64,160", "128,128", "128,160", "256,128", "256,160"]
    if "--tile" in sys.argv:    # auto against one forced tile
        tiles = [None, sys.argv[sys.argv.index("--tile") + 1]]
    if "--splits" in sys.argv:  # split-K variants (only honoured for small-M cases, where ops lends the workspace)
        tiles = [None, "64,128", "64,128,2", "64,128,3", "128,128,2", "128,128,3", "128,128,4", "128,160,2", "128,160,4",
                 "128,160,6", "256,160,4", "256,160,8"]
    for name, (H, W, c0, c1), N, ks, stride, up, glu in CASES:
        if sel and not any(s in name for s in sel):
            continue
        a0 = torch.randn(B, H, W, c0, device=dev, generator=g).bfloat16()
        a1 = torch.randn(B, H, W, c1, device=dev, generator=g).bfloat16() if c1 else None
        K = ks * ks * (c0 + c1)
        w = (torch.randn(N, K, device=dev, generator=g) / math.sqrt(K)).bfloat16()
        bias = torch.randn(N, device=dev, generator=g)
        if "--ones" in sys.argv:    # constant operands: the matrix cores toggle almost no bits -> what the schedule is worth without the power limit
            a0.fill_(1.0)
            w.fill_(1.0 / K)
            if a1 is not None:
                a1.fill_(1.0)
        kw = dict(a1=a1, ksize=ks, stride=stride, upsample=up, bias=bias, glu=glu)
        variants = [(t, None) for t in tiles]
        if "--pp" in sys.argv:      # the plain ring against its ping-pong form (LDM_GEMM_PINGPONG), same process, same buffers
            variants = [(t, pp) for t in tiles for pp in ("0", "2")]
        for tile, pp in variants:
            if pp is not None:
                os.environ["LDM_GEMM_PINGPONG"] = pp
            if tile is None:
                os.environ.pop("LDM_GEMM_TILE", None)
            elif glu and tile.split(",")[1] == "160":
                continue
            else:
                os.environ["LDM_GEMM_TILE"] = tile
            out = ops.conv_gemm(a0, w, **kw)
            M = out.numel() // out.shape[-1]
            for _ in range(3):
                ops.conv_gemm(a0, w, **kw)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                ops.conv_gemm(a0, w, **kw)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / iters
            if "--gap" in sys.argv:  # single launches after an idle gap (the clock has recovered): median of 7
                import time
                one = []
                for _ in range(7):
                    torch.cuda.synchronize()
                    time.sleep(0.03)
                    e0.record()
                    ops.conv_gemm(a0, w, **kw)
                    e1.record()
                    torch.cuda.synchronize()
                    one.append(e0.elapsed_time(e1) * 1e3)
                us = sorted(one)[3]
            fl = 2.0 * M * N * K
            print(f"{name:28s} M={M:6d} N={N:5d} K={K:5d} tile={tile or 'auto':8s}{' pp=' + pp if pp is not None else ''} {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s",
                  flush=True)
        os.environ.pop("LDM_GEMM_TILE", None)


if __name__ == "__main__":
    main()
