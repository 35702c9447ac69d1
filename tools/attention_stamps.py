"""Phase timing inside the bf16 attention kernel from in-kernel cycle stamps (diagnostic build: `make -C
latent-diffusion_amd/csrc stamps`), on the UNet's self-attention shapes.  Usage: python tools/attention_stamps.py"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion_amd"))
from ldm import _lib  # noqa: E402

_lib._LIB_PATH = os.path.join(ROOT, "latent-diffusion_amd", "lib", "libldm_amd_stamps.so")
from ldm import ops  # noqa: E402

CASES = [("L0 self 4096x4096 d40", 32, 4096, 4096, 8, 40), ("L1 self 1024x1024 d80", 32, 1024, 1024, 8, 80),
         ("L2 self 256x256 d160", 32, 256, 256, 8, 160), ("L0 cross 4096x77 d40", 32, 4096, 77, 8, 40)]
NAMES = ["prologue", "t8:wait-prev", "t8:commit+sync", "t8:QK", "t8:softmax", "t8:PV"]


def main():
    lib = _lib.load()
    lib.ldm_debug_set_attention_stamps.argtypes = [C.c_void_p, C.c_int64]
    lib.ldm_debug_set_attention_stamps.restype = C.c_int
    dev = torch.device("cuda")
    cap = 1 << 16
    buf = torch.zeros(cap * 16, dtype=torch.int64, device=dev)
    g = torch.Generator(device=dev).manual_seed(0)
    for name, B, lq, lk, H, d in CASES:
        q = torch.randn(B, lq, H * d, device=dev, generator=g).bfloat16()
        kv = torch.randn(B, lk, 2 * H * d, device=dev, generator=g).bfloat16()
        for _ in range(3):
            ops.attention(q, kv, H)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            ops.attention(q, kv, H)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 5
        lib.ldm_debug_set_attention_stamps(buf.data_ptr(), cap)
        buf.zero_()
        ops.attention(q, kv, H)
        torch.cuda.synchronize()
        lib.ldm_debug_set_attention_stamps(None, 0)
        st = buf.cpu().numpy().reshape(cap, 16)
        st = st[st[:, 0] != 0]
        fl = 4.0 * B * H * lq * lk * d
        line = f"{name:24s} {us:8.1f} us {fl / us / 1e6:7.1f} TFLOP/s  blocks {len(st):5d}  total {np.median(st[:, 8] - st[:, 0]):9.0f} cyc"
        line += f"  prologue {np.median(st[:, 1] - st[:, 0]):6.0f}"
        if (st[:, 7] != 0).all():
            d8 = np.median(np.diff(st[:, 2:8], axis=1).astype(np.float64), axis=0)
            line += "  | tile 8: " + "  ".join(f"{n.split(':')[1]} {v:5.0f}" for n, v in zip(NAMES[1:], d8))
            line += f"  | cyc/tile {np.median(st[:, 8] - st[:, 1]) / ((lk + 63) // 64):6.0f}"
        print(line, flush=True)


if __name__ == "__main__":
    main()
