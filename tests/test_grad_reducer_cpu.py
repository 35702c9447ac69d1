"""The N>1 training path on CPU: world_size-2 gloo processes feed ldm/parallel.GradReducer -- the bucketed gradient
all-reduce the HIP backward drives through Tape.put -- with different per-rank gradients in a fixed
arrival order.  Host logic only (bucket layout from the first pass, launch when a bucket fills, gradient accumulation
under no_sync, a parameter without a gradient, stray parameters); the result must be the mean over ranks."""
import os
import sys
import zlib

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "latent-diffusion_amd"))
sys.path.insert(0, ROOT)

SHAPES = {"a.weight": (40, 30), "a.bias": (40,), "b.weight": (64, 3, 3, 3), "c.weight": (1000,), "d.weight": (7, 5), "e.bias": (3,)}
ORDER = ["e.bias", "d.weight", "c.weight", "b.weight", "a.bias", "a.weight"]     # backward order: last layer first


def _grad(name, rank, step):
    g = torch.Generator().manual_seed(zlib.crc32(f"{name}/{rank}/{step}".encode()))
    return torch.randn(SHAPES[name], generator=g)


def _mean(name, world, step):
    return sum(_grad(name, r, step) for r in range(world)) / world


def _run_passes(rank, world):
    from ldm.parallel import GradReducer
    params = {k: torch.nn.Parameter(torch.zeros(s)) for k, s in SHAPES.items()}
    red = GradReducer(params, bucket_bytes=4096)          # small buckets: several collectives per pass
    # pass 0: layout is recorded, reduction happens at the end
    red.begin()
    out = {k: red.put(k, _grad(k, rank, 0)) for k in ORDER}
    out.update(red.finish())
    assert len(red.buckets) >= 3 and red.launched == len(red.buckets)
    for k in ORDER:
        assert torch.allclose(out[k], _mean(k, world, 0), atol=1e-6), k
    # pass 1: overlapped form; buckets launch as they fill
    red.begin()
    out, launched_mid = {}, None
    for i, k in enumerate(ORDER):
        out[k] = red.put(k, _grad(k, rank, 1))
        if i == 3:
            launched_mid = red.launched
    assert launched_mid >= 1, "no bucket was launched before the pass ended"
    assert red.finish() == {}
    for k in ORDER:
        assert torch.allclose(out[k], _mean(k, world, 1), atol=1e-6), k
    # pass 2: gradient accumulation (a micro-batch already sits in .grad), one parameter without a gradient
    acc = {k: _grad(k, rank, 7) for k in ORDER}
    for k in ORDER:
        params[k].grad = acc[k].clone()
    red.begin()
    out = {k: red.put(k, _grad(k, rank, 2)) for k in ORDER if k != "c.weight"}
    red.finish()
    for k in out:
        assert params[k].grad is None
        assert torch.allclose(out[k], _mean(k, world, 2) + _mean(k, world, 7), atol=1e-6), k
    assert params["c.weight"].grad is not None            # untouched: it took no part in this backward
    params["c.weight"].grad = None
    # pass 3: zero-copy form -- producers write into the bucket slices (dest), put() recognises them and copies nothing;
    # a strided gradient (the GLU row permutation) is copied once, straight into its slice
    red.begin()
    out = {}
    for k in ORDER:
        g = _grad(k, rank, 3)
        if k == "a.weight":                                # arrives as a transposed view with the parameter's element order
            out[k] = red.put(k, g.view(5, 8, 30).transpose(0, 1).contiguous().transpose(0, 1))
            continue
        d = red.dest(k, g.shape)
        assert d is not None and d.data_ptr() == red._view(k).data_ptr()
        d.copy_(g)
        out[k] = red.put(k, d)
        assert out[k].data_ptr() == d.data_ptr()
    red.finish()
    for k in ORDER:
        assert red.owns(k, out[k])
        assert torch.allclose(out[k], _mean(k, world, 3), atol=1e-6), k
        params[k].grad = out[k]                            # what the autograd boundary does (gradient_as_bucket_view)
    # pass 4: the gradients were not cleared and ARE the bucket slices: dest() refuses, put() accumulates in place
    red.begin()
    for k in ORDER:
        assert red.dest(k, SHAPES[k]) is None
        out[k] = red.put(k, _grad(k, rank, 4))
    red.finish()
    for k in ORDER:
        assert torch.allclose(out[k], _mean(k, world, 3) + _mean(k, world, 4), atol=1e-6), k
    return True


def _worker(rank, world, port):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert _run_passes(rank, world)
    finally:
        dist.destroy_process_group()


def test_single_process_reducer_is_identity():
    assert _run_passes(0, 1)


def test_two_gloo_ranks_average_gradients():
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port), nprocs=2, join=True)


def test_stray_parameter_after_layout_is_fixed():
    from ldm.parallel import GradReducer
    params = {k: torch.nn.Parameter(torch.zeros(s)) for k, s in SHAPES.items()}
    red = GradReducer(params, bucket_bytes=4096)
    red.begin()
    for k in ORDER[1:]:
        red.put(k, _grad(k, 0, 0))
    red.finish()
    red.begin()
    out = {k: red.put(k, _grad(k, 0, 1)) for k in ORDER}   # e.bias was not in the first pass
    red.finish()
    for k in ORDER:
        assert torch.allclose(out[k], _grad(k, 0, 1), atol=1e-6)
