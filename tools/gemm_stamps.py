"""Phase timing inside the bf16 implicit-GEMM kernel from in-kernel cycle stamps (diagnostic build: `make -C
latent-diffusion_amd/csrc stamps`).  Per workgroup: setup, first-tile fill, main loop, epilogue (shader cycles, median
over workgroups), plus the number of workgroups in flight per CU derived from the 100 MHz real-time stamps.
Usage: python tools/gemm_stamps.py [case-substring ...]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion_amd"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
from ldm import _lib  # noqa: E402

_lib._LIB_PATH = os.path.join(ROOT, "latent-diffusion_amd", "lib", "libldm_amd_stamps.so")
from ldm import ops  # noqa: E402
from gemm_bench import B, CASES  # noqa: E402
import math  # noqa: E402


def main():
    sel = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = _lib.load()
    lib.ldm_debug_set_stamps.argtypes = [C.c_void_p, C.c_int64]
    lib.ldm_debug_set_stamps.restype = C.c_int
    dev = torch.device("cuda")
    cap = 1 << 17
    buf = torch.zeros(cap * 16, dtype=torch.int64, device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    print(f"{'case':28s} {'tile':>10s} {'blocks':>6s} | {'setup':>6s} {'fill':>6s} {'loop':>7s} {'cyc/kt':>6s} {'epi':>6s} {'total':>7s} | "
          f"{'us':>7s} {'wg/CU':>5s} {'clkGHz':>6s}")
    print("kt8 columns: 2-stage tiles = stage issue, math, barrier(+DMA wait); 3-slot ring = barrier, stage issue, math")
    for name, (H, W, c0, c1), N, ks, stride, up, glu in CASES:
        if sel and not any(s in name for s in sel):
            continue
        a0 = torch.randn(B, H, W, c0, device=dev, generator=g).bfloat16()
        a1 = torch.randn(B, H, W, c1, device=dev, generator=g).bfloat16() if c1 else None
        K = ks * ks * (c0 + c1)
        w = (torch.randn(N, K, device=dev, generator=g) / math.sqrt(K)).bfloat16()
        bias = torch.randn(N, device=dev, generator=g)
        kw = dict(a1=a1, ksize=ks, stride=stride, upsample=up, bias=bias, glu=glu)
        log = []
        ops.set_launch_log(log)
        for _ in range(5):
            ops.conv_gemm(a0, w, **kw)
        lib.ldm_debug_set_stamps(buf.data_ptr(), cap)
        buf.zero_()
        ops.conv_gemm(a0, w, **kw)
        torch.cuda.synchronize()
        lib.ldm_debug_set_stamps(None, 0)
        ops.set_launch_log(None)
        tile = log[-1][0][log[-1][0].index("<"):]
        st = buf.cpu().numpy().reshape(cap, 16)
        st = st[st[:, 0] != 0]
        if len(st) == 0:
            print(f"{name:28s} {tile:>10s}  (too many workgroups for the stamp buffer)")
            continue
        d = np.diff(st[:, :5], axis=1).astype(np.float64)
        med = np.median(d, axis=0)
        tot = np.median(st[:, 4] - st[:, 0])
        bk = int(tile.strip("<>").split(",")[2])
        span = (st[:, 6].max() - st[:, 5].min()) * 10e-9            # seconds (100 MHz)
        busy = ((st[:, 6] - st[:, 5]) * 10e-9).sum()
        clk = np.median((st[:, 4] - st[:, 0]) / np.maximum((st[:, 6] - st[:, 5]) * 10e-9, 1e-9)) / 1e9
        inner = ""
        if (st[:, 8] != 0).all():   # K-tile 8 of the main loop: the four sub-phases between stamps 8..11
            di = np.median(np.diff(st[:, 8:12], axis=1).astype(np.float64), axis=0)
            inner = f" | kt8: {di[0]:5.0f} {di[1]:5.0f} {di[2]:5.0f}"
        if (st[:, 12] != 0).all() and (st[:, 2] != 0).all():   # ping-pong tiles: starts of K-tiles 1..3 after the first-tile barrier (stamp 2)
            pts = [st[:, 2]] + [st[:, c] for c in (12, 13, 14) if (st[:, c] != 0).all()]
            dd = [np.median((b_ - a_).astype(np.float64)) for a_, b_ in zip(pts[:-1], pts[1:])]
            inner += " | first K-tiles: " + " ".join(f"{v:5.0f}" for v in dd)
        print(f"{name:28s} {tile:>10s} {len(st):6d} | {med[0]:6.0f} {med[1]:6.0f} {med[2]:7.0f} {med[2] / (K // bk):6.0f} {med[3]:6.0f} "
              f"{tot:7.0f} | {span * 1e6:7.1f} {busy / span / 256:5.2f} {clk:6.2f}{inner}", flush=True)


if __name__ == "__main__":
    main()
