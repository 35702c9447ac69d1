#!/usr/bin/env python3
"""VAE-decoder leg of `generate` (train_latent_diffusion.py:596-601): latents -> uint8 images, bf16.

    python tools/vae_bench.py --batch 8 --latent 64 --steps 5

Prints one JSON line: images/s, ms per batch, the decoder's FLOPs per image (convs + attention GEMMs) and the fraction of
the dense bf16 MFMA peak that is.  SD-1.x decoder, random weights, synthetic latents."""
from __future__ import annotations

import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion_amd"))
sys.path.insert(0, ROOT)

from bench import MFMA_BF16_DENSE_PEAK  # noqa: E402


def decoder_flops(cfg, h, w):
    """2 * MACs of every conv / linear / attention GEMM for one latent of h x w."""
    rev = list(cfg.block_out_channels)[::-1]
    top = rev[0]
    pix = h * w
    f = 2 * pix * (cfg.latent_channels ** 2 + 9 * cfg.latent_channels * top)

    def resnet(ci, co, p):
        return 2 * p * (9 * ci * co + 9 * co * co + (ci * co if ci != co else 0))
    f += 2 * resnet(top, top, pix)
    f += 2 * pix * 4 * top * top + 4 * pix * pix * top            # q, k, v, out projections; q k^T and p v
    prev = top
    for i, c in enumerate(rev):
        for j in range(cfg.layers_per_block + 1):
            f += resnet(prev if j == 0 else c, c, pix)
        if i != len(rev) - 1:
            pix *= 4
            f += 2 * pix * 9 * c * c
        prev = c
    f += 2 * pix * 9 * rev[-1] * cfg.out_channels
    return f


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--latent", type=int, default=64)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    args = ap.parse_args()

    from bench import init_weights_
    from ldm.vae import AutoencoderKL, to_uint8

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    with torch.device("meta"):
        vae = AutoencoderKL()
    vae = vae.to_empty(device=dev)
    init_weights_(vae, seed=77)
    vae.eval()
    lat = torch.randn(args.batch, 4, args.latent, args.latent, device=dev)

    def step():
        with torch.autocast("cuda", dtype=torch.bfloat16):
            img = vae.decode(lat / vae.config.scaling_factor, return_dict=False)[0]
        return to_uint8(img)

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.steps):
        out = step()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.steps
    fl = decoder_flops(vae.config, args.latent, args.latent)
    assert out.dtype == torch.uint8 and out.shape == (args.batch, 3, 8 * args.latent, 8 * args.latent)
    print(json.dumps({"metric": "VAE decode images/s (latents -> uint8 images)", "value": round(args.batch / ms * 1e3, 2),
                      "unit": "img/s", "ms_per_batch": round(ms, 2), "batch": args.batch,
                      "latent": [args.latent, args.latent], "image": [8 * args.latent, 8 * args.latent], "dtype": "bf16",
                      "gflop_per_image": round(fl / 1e9, 1),
                      "frac_of_mfma_peak": round(fl * args.batch / (ms * 1e-3) / MFMA_BF16_DENSE_PEAK, 4),
                      "peak_device_memory_gb": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 2),
                      "data": "synthetic"}), flush=True)


if __name__ == "__main__":
    main()
