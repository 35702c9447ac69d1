"""GPU parity of the fused training-step tail (ldm/optim.py over ldm_mse_loss / ldm_grad_norm_clip / ldm_adamw_step)
against the torch ops the reference's trainer uses (train_latent_diffusion.py:1108, :1123, :909-915): float32, so the
gate is a few ulps -- the arithmetic is the same, only the reduction order and one reciprocal differ."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2, seeded

pytestmark = pytest.mark.gpu

SHAPES = [(320, 4, 3, 3), (320,), (1280, 320), (7,), (100000,), (33, 17, 3, 3), (1,), (16384,), (16385,)]


def _params(seed):
    return [torch.nn.Parameter(seeded(s, seed + i).cuda()) for i, s in enumerate(SHAPES)]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_mse_loss_and_gradient(dtype):
    from ldm.optim import mse_loss
    pred = seeded((4, 4, 32, 32), 1).cuda().to(dtype).requires_grad_(True)
    noise = seeded((4, 4, 32, 32), 2).cuda()
    ref = F.mse_loss(pred.float(), noise.float(), reduction="mean")
    ref.backward()
    loss, dpred = mse_loss(pred.detach(), noise)
    assert loss.dtype == torch.float32 and loss.dim() == 0
    assert abs(float(loss) - float(ref)) < 2e-6 * float(ref)
    assert dpred.dtype == dtype and rel_l2(dpred, pred.grad) < (1e-6 if dtype == torch.float32 else 4e-3)
    loss2, none = mse_loss(pred.detach()[:1, :1, :3, :5], noise[:1, :1, :3, :5], want_grad=False)   # ragged, tiny
    assert none is None and abs(float(loss2) - float(F.mse_loss(pred.detach()[:1, :1, :3, :5].float(), noise[:1, :1, :3, :5]))) < 1e-6


def test_clip_grad_norm_matches_torch():
    from ldm.optim import clip_grad_norm_
    ps, qs = _params(10), _params(10)
    for i, (p, q) in enumerate(zip(ps, qs)):
        g = seeded(tuple(p.shape), 50 + i).cuda() * 3.0
        p.grad, q.grad = g.clone(), g.clone()
    n_ref = torch.nn.utils.clip_grad_norm_(qs, 1.0)
    n = clip_grad_norm_(ps, 1.0)
    assert abs(float(n) - float(n_ref)) < 1e-5 * float(n_ref)
    for p, q in zip(ps, qs):
        assert rel_l2(p.grad, q.grad) < 1e-6
    n2 = clip_grad_norm_(ps, 1e9)                      # norm below the limit: gradients untouched
    assert float(n2) <= 1.0 + 1e-5
    for p, q in zip(ps, qs):
        assert rel_l2(p.grad, q.grad) < 1e-6


@pytest.mark.parametrize("clip", [None, 1.0])
def test_fused_adamw_matches_torch(clip):
    from ldm.optim import FusedAdamW
    ps, qs = _params(20), _params(20)
    kw = dict(lr=1e-3, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    fused = FusedAdamW(ps, max_grad_norm=clip, **kw)
    ref = torch.optim.AdamW(qs, **kw)
    sched = torch.optim.lr_scheduler.LambdaLR(fused, lambda s: 1.0 / (1 + s))      # param_groups interop
    sched_ref = torch.optim.lr_scheduler.LambdaLR(ref, lambda s: 1.0 / (1 + s))
    for it in range(5):
        for i, (p, q) in enumerate(zip(ps, qs)):
            g = seeded(tuple(p.shape), 100 * it + i).cuda() * (2.0 if it % 2 else 0.01)
            p.grad, q.grad = g.clone(), g.clone()
        v0 = ps[0]._version
        if clip is not None:
            n_ref = torch.nn.utils.clip_grad_norm_(qs, clip)
        fused.step()
        ref.step()
        sched.step()
        sched_ref.step()
        assert ps[0]._version > v0                     # packed-weight caches key on the version counter
        if clip is not None:
            assert abs(float(fused.last_grad_norm) - float(n_ref)) < 1e-5 * float(n_ref)
        for p, q in zip(ps, qs):
            assert rel_l2(p, q) < 2e-6, it
    sd = fused.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    for k in range(len(ps)):
        assert rel_l2(sd["state"][k]["exp_avg_sq"], ref.state_dict()["state"][k]["exp_avg_sq"]) < 2e-6


def test_fused_tail_trains_the_small_unet():
    """The fused tail around the HIP UNet: three steps of the reference's train step with mse_loss / FusedAdamW(clip) against
    the same steps with F.mse_loss / clip_grad_norm_ / torch AdamW (same kernels for the UNet, float32)."""
    from ldm.optim import FusedAdamW, mse_loss
    from ldm.unet import UNet
    from oracle import unet_ref as uref
    kw = dict(block_out_channels=[32, 64, 64, 64])
    sd = uref.fill_state_dict(uref.param_shapes(uref.UNetConfig(**kw)), seed=7, gain=1.0)
    nets = []
    for _ in range(2):
        n = UNet(**kw)
        n.load_state_dict(sd)
        nets.append(n.cuda().train())
    a, b = nets
    oa = FusedAdamW(a.parameters(), lr=1e-4, max_grad_norm=1.0)
    ob = torch.optim.AdamW(b.parameters(), lr=1e-4, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    x, ctx = seeded((2, 4, 16, 16), 11).cuda(), seeded((2, 77, 768), 12).cuda()
    t = torch.tensor([17, 801], device="cuda")
    for it in range(3):
        noise = seeded((2, 4, 16, 16), 30 + it).cuda()
        pa = a(x, t, ctx)
        la, dpa = mse_loss(pa.detach(), noise)
        pa.backward(dpa)
        oa.step()
        oa.zero_grad(set_to_none=True)
        pb = b(x, t, ctx)
        lb = F.mse_loss(pb.float(), noise.float())
        lb.backward()
        torch.nn.utils.clip_grad_norm_(b.parameters(), 1.0)
        ob.step()
        ob.zero_grad(set_to_none=True)
        assert abs(float(la) - float(lb)) < 1e-4 * float(lb), it
    # Adam's first updates are +-lr whatever the gradient's size, so parameters whose gradient is mathematically zero
    # (rounding noise of either sign, see test_backward_gpu.py) may move apart by 2 lr; all parameters together must agree
    worst = max(rel_l2(p, q) for p, q in zip(a.parameters(), b.parameters()))
    flat = lambda n: torch.cat([p.detach().flatten() for p in n.parameters()])
    total = rel_l2(flat(a), flat(b))
    print(f"parameters after 3 fused-tail steps vs torch tail: all together {total:.2e}, worst single tensor {worst:.2e}")
    assert total < 2e-5 and worst < 5e-3


@pytest.mark.parametrize("make", ["fused_torch", "foreach_torch", "ldm_fused"])
def test_packed_weights_follow_every_optimizer(make):
    """The derived (packed) filters must be rebuilt after any optimizer step.  torch.optim.AdamW(fused=True) updates the
    parameters without bumping their version counters: the cache keys on a global optimizer-step count as well."""
    from ldm.optim import FusedAdamW
    from ldm.unet import UNet
    from oracle import unet_ref as uref
    kw = dict(block_out_channels=[32, 64, 64, 64])
    sd = uref.fill_state_dict(uref.param_shapes(uref.UNetConfig(**kw)), seed=7, gain=1.0)
    net = UNet(**kw)
    net.load_state_dict(sd)
    net = net.cuda()
    x, ctx = seeded((2, 4, 16, 16), 11).cuda(), seeded((2, 77, 768), 12).cuda()
    with torch.no_grad():
        y0 = net(x, 10, ctx)
    opt = {"fused_torch": lambda p: torch.optim.AdamW(p, lr=1e-2, fused=True),
           "foreach_torch": lambda p: torch.optim.AdamW(p, lr=1e-2, foreach=True),
           "ldm_fused": lambda p: FusedAdamW(p, lr=1e-2)}[make](net.parameters())
    for i, p in enumerate(net.parameters()):
        p.grad = seeded(tuple(p.shape), 200 + i).cuda()
    opt.step()
    with torch.no_grad():
        y1 = net(x, 10, ctx)
    fresh = UNet(**kw)
    fresh.load_state_dict(net.state_dict())
    fresh = fresh.cuda()
    with torch.no_grad():
        y2 = fresh(x, 10, ctx)
    assert not torch.equal(y0, y1)
    assert torch.equal(y1, y2)


def test_channels_last_module_trains_with_every_optimizer():
    """The reference trainer calls diffusion_model.to(memory_format=torch.channels_last) (train_latent_diffusion.py:876)
    before building the optimizer.  One train step after that call with torch AdamW, FusedAdamW and FusedAdamW over
    DataParallel's bucket-view gradients must equal the same step on a module that was never re-laid; and FusedAdamW /
    clip_grad_norm_ on genuinely channels_last parameters and gradients (dense, non-contiguous) must match torch."""
    from ldm.optim import FusedAdamW, clip_grad_norm_
    from ldm.parallel import DataParallel
    from ldm.unet import UNet
    from oracle import unet_ref as uref
    kw = dict(block_out_channels=[32, 64, 64, 64])
    sd = uref.fill_state_dict(uref.param_shapes(uref.UNetConfig(**kw)), seed=7, gain=1.0)
    x, ctx = seeded((2, 4, 16, 16), 11).cuda(), seeded((2, 77, 768), 12).cuda()
    t = torch.tensor([17, 801], device="cuda")
    noise = seeded((2, 4, 16, 16), 30).cuda()

    def one_step(make_opt, channels_last, wrap=False):
        net = UNet(**kw)
        net.load_state_dict(sd)
        net = net.cuda().train()
        if channels_last:
            net = net.to(memory_format=torch.channels_last)
        model = DataParallel(net) if wrap else net
        opt = make_opt(net.parameters())
        F.mse_loss(model(x, t, ctx).float(), noise).backward()
        if not isinstance(opt, FusedAdamW):
            torch.nn.utils.clip_grad_norm_(net.parameters(), 1.0)
        opt.step()
        return torch.cat([p.detach().flatten() for p in net.parameters()])

    adamw = lambda p: torch.optim.AdamW(p, lr=1e-4, betas=(0.9, 0.999), weight_decay=1e-2, eps=1e-8)
    fused = lambda p: FusedAdamW(p, lr=1e-4, max_grad_norm=1.0)
    base = one_step(adamw, False)
    assert torch.equal(one_step(adamw, True), base)
    f0 = one_step(fused, False)
    assert rel_l2(f0, base) < 2e-5
    assert torch.equal(one_step(fused, True), f0)
    assert torch.equal(one_step(fused, True, wrap=True), f0)

    # dense, non-contiguous tensors through the multi-tensor kernels
    shapes = [(64, 32, 3, 3), (48, 16, 3, 3), (40,), (24, 8, 1, 1)]
    ps, qs = [], []
    for i, shp in enumerate(shapes):
        v = seeded(shp, 70 + i).cuda()
        if len(shp) == 4:
            v = v.contiguous(memory_format=torch.channels_last)
        ps.append(torch.nn.Parameter(v.clone(memory_format=torch.preserve_format)))
        qs.append(torch.nn.Parameter(v.clone(memory_format=torch.preserve_format)))
    assert not ps[0].is_contiguous()
    of, ot = FusedAdamW(ps, lr=1e-4, max_grad_norm=1.0), adamw(qs)
    for it in range(3):
        for i, (p, q) in enumerate(zip(ps, qs)):
            g = seeded(tuple(p.shape), 90 + 10 * it + i).cuda() * 2.0
            # gradient layouts a trainer can produce: the parameter's own strides, or plain contiguous
            p.grad = g.contiguous(memory_format=torch.channels_last) if (p.dim() == 4 and it % 2 == 0) else g.clone()
            q.grad = g.clone()
        torch.nn.utils.clip_grad_norm_(qs, 1.0)
        of.step()
        ot.step()
    for p, q in zip(ps, qs):
        assert p.stride() == q.stride() and rel_l2(p, q) < 1e-6
    for p, q in zip(ps, qs):
        p.grad = q.grad = None
        g = seeded(tuple(p.shape), 300).cuda() * 5
        p.grad = g.contiguous(memory_format=torch.channels_last) if p.dim() == 4 else g.clone()
        q.grad = g.clone()
    n, n_ref = clip_grad_norm_(ps, 1.0), torch.nn.utils.clip_grad_norm_(qs, 1.0)
    assert abs(float(n) - float(n_ref)) < 1e-5 * float(n_ref)
    for p, q in zip(ps, qs):
        assert rel_l2(p.grad, q.grad) < 1e-6
