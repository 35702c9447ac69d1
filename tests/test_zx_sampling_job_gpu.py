"""BASELINE.json configs[2] as a -m gpu test at its real size: 256 latents of 4x64x64, 50 reverse steps, default UNet
(859.5 M parameters, random init) under bf16 autocast, batch-sharded over 8 ranks of 32 latents (here the 8 ranks run one
after the other on the one GPU of the test box), no data-path collective.  Collected late on purpose (it is the longest
GPU test: two passes over the 12,800 img-step job).

What can be checked at this size without an oracle that would run for hours: every latent is independent through all steps
and its noise is a function of (seed, step, global sample index), so the eight 32-latent shards must reproduce the
single-process job run in micro-batches of 32 BIT FOR BIT (same chunks, same kernels, deterministic kernels); the result is
finite and the job's rate is printed.  Parity of the forward and of the step at this shape against the oracle is
tests/test_unet_gpu.py::test_default_unet_64x64_headline_shape and tests/test_scheduler_gpu.py.  The same function runs in
miniature on the host emulator (tests/emu: `python tests/test_zx_sampling_job_gpu.py emu`)."""
import os
import sys
import time

import pytest
import torch

from conftest import same_bits, seeded


def run_job_and_shards(net, make_scheduler, ctx, *, latent, steps, micro, world, autocast_bf16, device):
    """(single-process job in micro-batches, the same job as `world` shards, seconds of the first) -- ldm/sampling.py."""
    from ldm import sampling
    kw = dict(latent_shape=(4, latent, latent), num_inference_steps=steps, seed=17, micro_batch=micro, device=device)

    def call(**extra):
        if autocast_bf16:
            with torch.autocast("cuda", dtype=torch.bfloat16):
                return sampling.sample_latents(net, make_scheduler(), ctx, **kw, **extra)
        return sampling.sample_latents(net, make_scheduler(), ctx, **kw, **extra)

    t0 = time.perf_counter()
    whole = call()
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    seconds = time.perf_counter() - t0
    parts = [call(rank=r, world=world) for r in range(world)]
    return whole, torch.cat(parts, dim=0), seconds


@pytest.mark.gpu
def test_config2_job_256_latents_50_steps_sharded_over_8():
    from ldm.scheduling_ddpm import DDPMScheduler
    from ldm.unet import UNet
    from oracle import unet_ref as uref
    dev = torch.device("cuda", torch.cuda.current_device())
    cfg = uref.UNetConfig()
    net = UNet()
    net.load_state_dict(uref.fill_state_dict(uref.param_shapes(cfg), seed=7, **uref.WELL_CONDITIONED))
    net = net.to(dev).eval()
    n, steps = 256, 50
    ctx = seeded((n, 77, 768), 31).to(dev)
    whole, sharded, seconds = run_job_and_shards(net, lambda: DDPMScheduler(device=dev), ctx, latent=64, steps=steps, micro=32,
                                                 world=8, autocast_bf16=True, device=dev)
    assert whole.shape == (n, 4, 64, 64) and whole.dtype == torch.float32
    assert bool(torch.isfinite(whole).all())
    assert same_bits(sharded, whole), "8 shards of 32 latents differ from the single-process job"
    assert float(whole.std()) > 1e-3       # not a constant
    print(f"configs[2] job on one GPU: {n * steps} img-steps in {seconds:.1f} s = {n * steps / seconds:.0f} img-steps/s")


def _emulated_miniature():
    """The same function on the host emulator: small UNet, 6 latents of 4x8x8, 2 steps, micro-batches of 2, 3 shards."""
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "emu"))
    from harness import emulated
    from ldm.scheduling_ddpm import DDPMScheduler
    from ldm.unet import UNet
    from oracle import unet_ref as uref
    kwn = dict(block_out_channels=[32, 64, 64, 64])
    net = UNet(**kwn)
    net.load_state_dict(uref.fill_state_dict(uref.param_shapes(uref.UNetConfig(**kwn)), seed=7, **uref.WELL_CONDITIONED))
    net.eval()
    dev = torch.device("cpu")
    with emulated("lazy", guard=True):
        ctx = seeded((6, 77, 768), 31)
        whole, sharded, seconds = run_job_and_shards(net, lambda: DDPMScheduler(device=dev), ctx, latent=8, steps=2, micro=2,
                                                     world=3, autocast_bf16=False, device=dev)
        assert whole.shape == (6, 4, 8, 8) and bool(torch.isfinite(whole).all())
        assert same_bits(sharded, whole)
        print(f"emulated miniature: shards == job bit for bit ({seconds:.0f} s for the job)")


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "emu":
        _emulated_miniature()
