#!/usr/bin/env python3
"""Time of re-deriving every packed filter of the default UNet (what a training step pays after the optimizer step):
forward + input-gradient layouts, bf16.  `LDM_AMD_LIB=<other build> python tools/pack_bench.py` for A/B runs."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion_amd"))
sys.path.insert(0, ROOT)
from bench import init_weights_  # noqa: E402
from ldm.unet import UNet, pack_tree  # noqa: E402

dev = torch.device("cuda", 0)
with torch.device("meta"):
    net = UNet()
net = net.to_empty(device=dev)
init_weights_(net, seed=1)
for train in (False, True):
    for _ in range(2):
        pack_tree(net, torch.bfloat16, train)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        pack_tree(net, torch.bfloat16, train)
    e1.record()
    torch.cuda.synchronize()
    print(f"pack_tree(train={train}): {e0.elapsed_time(e1) / 5:.2f} ms  lib={os.environ.get('LDM_AMD_LIB', 'in-tree')}", flush=True)
