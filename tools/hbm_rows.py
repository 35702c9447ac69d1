"""The four memory-bound rows of bench.py's `config.hbm_kernels` on their own (scheduler step, add_noise, GroupNorm + SiLU,
LayerNorm on DRAM-resident, rotated operand sets), so that rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE can be put beside them:
tools/hbm_rows_pmc.sh.  Usage: python tools/hbm_rows.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion_amd"))
import bench  # noqa: E402
from ldm import ops  # noqa: E402
from ldm.scheduling_ddpm import DDPMScheduler  # noqa: E402


def main():
    dev = torch.device("cuda")
    sched = DDPMScheduler(device=dev)
    sched.set_timesteps(50)
    for r in bench.hbm_kernel_rows(dev, sched, ops):
        print(json.dumps(r))


if __name__ == "__main__":
    main()
