"""Child process of tests/test_parallel_gpu.py (never collected by pytest).  Started as `python tests/_dist_child.py MODE`
with RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment, BEFORE anything has touched the GPU in this process.

  nccl1   one rank, backend "nccl" (= RCCL): the process group is created first, bound to cuda:0; then DataParallel with
          always_reduce drives real RCCL all-reduces of the gradient buckets from inside the HIP backward
  gloo2   two ranks sharing cuda:0, backend gloo (device buckets staged through host memory: gloo's CUDA-tensor collectives
          hang on this platform): the mean over ranks of different per-rank batches
"""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion_amd"))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SMALL = dict(block_out_channels=[32, 64, 64, 64])


def seeded(shape, seed):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed))


def small_unet():
    from ldm.unet import UNet
    from oracle import unet_ref as uref
    net = UNet(**SMALL)
    net.load_state_dict(uref.fill_state_dict(uref.param_shapes(uref.UNetConfig(**SMALL)), seed=7, gain=1.0))
    return net.cuda().train()


def default_unet():
    from bench import init_weights_
    from ldm.unet import UNet
    with torch.device("meta"):
        net = UNet()
    net = net.to_empty(device="cuda")
    init_weights_(net, seed=5)
    return net.train()


def grads_of(net):
    return {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}


def three_passes(model, net, inputs, autocast):
    """sync, then no_sync + sync (gradient accumulation): returns the gradients after the first and after the third."""
    def fb(i):
        x, t, ctx, noise = inputs[i]
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=autocast):
            pred = model(x, t, ctx)
        torch.nn.functional.mse_loss(pred.float(), noise).backward()
    net.zero_grad(set_to_none=True)
    fb(0)
    first = grads_of(net)
    launched = getattr(getattr(model, "reducer", None), "launched", None)
    net.zero_grad(set_to_none=True)
    ctxm = model.no_sync() if hasattr(model, "no_sync") else torch.enable_grad()
    with ctxm:
        fb(1)
    fb(2)
    return first, grads_of(net), launched


def run_nccl1():
    from ldm import sampling
    from ldm.parallel import DataParallel
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", world_size=1, rank=0, device_id=dev)   # first GPU use of this process
    assert dist.get_backend() == "nccl"
    for name, make, B, hw, autocast, bucket_mb in (("small", small_unet, 2, 16, False, 1.0), ("default", default_unet, 2, 32, True, 64.0)):
        inputs = [(seeded((B, 4, hw, hw), 10 + i).cuda(), torch.tensor([17, 801][:B], device="cuda") + i,
                   seeded((B, 77, 768), 20 + i).cuda(), seeded((B, 4, hw, hw), 30 + i).cuda()) for i in range(3)]
        net = make()
        ref1, ref3, _ = three_passes(net, net, inputs, autocast)
        dp = DataParallel(net, bucket_mb=bucket_mb, always_reduce=True)
        assert dp.reducer.always_reduce
        for rep in range(2):           # rep 0: first pass records the bucket layout; rep 1: every pass overlapped
            got1, got3, launched = three_passes(dp, net, inputs, autocast)
            nb = len(dp.reducer.buckets)
            assert nb >= (3 if name == "small" else 40), nb
            assert launched == nb, (launched, nb)
            for k in ref1:
                assert torch.equal(got1[k], ref1[k]), (name, rep, "pass 1", k)
                assert torch.equal(got3[k], ref3[k]), (name, rep, "accumulated", k)
        torch.cuda.synchronize()
        print(f"nccl1 {name}: {nb} RCCL bucket all-reduces per backward, gradients bit-equal to the un-wrapped run", flush=True)
        del net, dp
        torch.cuda.empty_cache()
    lat = seeded((5, 4, 16, 16), 40).cuda()
    out = sampling.gather_latents(lat, 5, 0, 1, force=True)
    assert torch.equal(out, lat) and out.data_ptr() != lat.data_ptr()
    t = sampling.max_over_ranks(1.5, dev, 1)
    assert t == 1.5
    dist.barrier()
    dist.destroy_process_group()


def run_gloo2():
    from ldm.parallel import DataParallel
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)                        # both ranks share the one GPU of the box
    dist.init_process_group("gloo", world_size=world, rank=rank)
    print(f"gloo2 rank {rank}: group up", flush=True)
    net = small_unet()
    B = 2
    def batch(r, i):
        return (seeded((B, 4, 16, 16), 100 * r + i).cuda(), torch.tensor([17, 801], device="cuda") + i + r,
                seeded((B, 77, 768), 100 * r + 20 + i).cuda(), seeded((B, 4, 16, 16), 100 * r + 30 + i).cuda())
    # single-process reference: the mean over both ranks' batches
    want = []
    for i in range(2):
        acc = None
        for r in range(world):
            net.zero_grad(set_to_none=True)
            x, t, ctx, noise = batch(r, i)
            torch.nn.functional.mse_loss(net(x, t, ctx).float(), noise).backward()
            g = grads_of(net)
            acc = g if acc is None else {k: acc[k] + g[k] for k in g}
        want.append({k: v / world for k, v in acc.items()})
    net.zero_grad(set_to_none=True)
    print(f"gloo2 rank {rank}: reference gradients done", flush=True)
    dp = DataParallel(net, bucket_mb=1.0)
    print(f"gloo2 rank {rank}: weights broadcast", flush=True)
    for i in range(2):                              # layout pass, then overlapped pass
        net.zero_grad(set_to_none=True)
        x, t, ctx, noise = batch(rank, i)
        torch.nn.functional.mse_loss(dp(x, t, ctx).float(), noise).backward()
        assert dp.reducer.launched == len(dp.reducer.buckets) >= 3
        worst = 0.0
        for k, p in net.named_parameters():
            e = float((p.grad - want[i][k]).norm() / (want[i][k].norm() + 1e-20))
            worst = max(worst, e)
        assert worst < 1e-5, (i, worst)
        print(f"gloo2 rank {rank}: pass {i} ok", flush=True)
    torch.cuda.synchronize()
    dist.barrier()
    if rank == 0:
        print(f"gloo2: gradients of {world} ranks on one GPU = single-process mean (worst rel-L2 {worst:.1e})", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    import faulthandler
    faulthandler.dump_traceback_later(100, exit=True)      # a hung collective names its line instead of hanging the suite
    {"nccl1": run_nccl1, "gloo2": run_gloo2}[sys.argv[1]]()
