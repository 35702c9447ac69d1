"""Generate the golden fixtures in tests/golden/ by importing the REFERENCE in the build container.

Run once, here (python tests/golden/make_golden.py); the GPU box never sees /root/reference.
The reference has no tests or golden vectors of its own, so these vectors are
produced by running its code on CPU in fp32 with two inert stand-ins for absent third-party modules
(SURVEY.md appendix B): ``jaxtyping`` (annotation only) and
``diffusers.models.embeddings.get_timestep_embedding`` (restated published formula, "parity unpinned").

While generating, every vector is also compared with the oracle restatement (oracle/*.py); the script
aborts if they disagree, so a committed fixture certifies oracle == reference on that input.
Fixtures hold data only: seeds, small inputs, expected outputs.
"""
from __future__ import annotations

import hashlib
import json
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import scheduler_ref as sref  # noqa: E402
from oracle import unet_ref as uref  # noqa: E402


def _install_standins():
    def get_timestep_embedding(timesteps, embedding_dim, flip_sin_to_cos=False, downscale_freq_shift=1,
                               scale=1, max_period=10000):
        half = embedding_dim // 2
        exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
        exponent = exponent / (half - downscale_freq_shift)
        emb = scale * (timesteps[:, None].float() * torch.exp(exponent)[None, :])
        emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
        if flip_sin_to_cos:
            emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
        if embedding_dim % 2 == 1:
            emb = torch.nn.functional.pad(emb, (0, 1, 0, 0))
        return emb

    emb_mod = types.ModuleType("diffusers.models.embeddings")
    emb_mod.get_timestep_embedding = get_timestep_embedding
    sys.modules.update({"diffusers": types.ModuleType("diffusers"),
                        "diffusers.models": types.ModuleType("diffusers.models"),
                        "diffusers.models.embeddings": emb_mod})

    class _Sub:
        def __getitem__(self, _):
            return torch.Tensor

    jt = types.ModuleType("jaxtyping")
    jt.Float = _Sub()
    sys.modules["jaxtyping"] = jt


_install_standins()
sys.path.insert(0, "/root/reference")
from ldm import unet as R  # noqa: E402  (reference)
from ldm.scheduling_ddpm import DDPMScheduler as RefScheduler  # noqa: E402

CPU = torch.device("cpu")


def sha(t: torch.Tensor) -> str:
    return hashlib.sha256(t.detach().contiguous().numpy().tobytes()).hexdigest()[:16]


def seeded(shape, seed, dtype=torch.float32):
    return torch.randn(shape, generator=torch.Generator().manual_seed(seed), dtype=dtype)


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **{k: (v.detach().numpy() if torch.is_tensor(v) else np.asarray(v))
                                 for k, v in arrays.items()})
    print(f"  wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


# ----------------------------------------------------------------------------- scheduler
def gen_scheduler():
    print("scheduler")
    out = {}
    meta = {}
    for sched in ("linear", "cosine", "sigmoid"):
        ref = RefScheduler(beta_schedule=sched, device=CPU)
        mine = sref.build_tables(sched, 1000)
        for k in mine:
            r = getattr(ref, k)
            assert torch.equal(r, mine[k]), (sched, k)
            out[f"{sched}.{k}"] = r
            meta[f"{sched}.{k}"] = sha(r)
    for spacing in ("leading", "linspace", "trailing"):
        for n in (50, 10, 3, 1000, 7):
            ref = RefScheduler(timestep_spacing=spacing, device=CPU)
            ref.set_timesteps(n, device="cpu")
            mine = sref.inference_timesteps(n, 1000, spacing)
            assert torch.equal(ref.timesteps, mine), (spacing, n)
            out[f"timesteps.{spacing}.{n}"] = ref.timesteps
    ref = RefScheduler(timestep_spacing="leading", steps_offset=1, device=CPU)
    ref.set_timesteps(20)
    assert torch.equal(ref.timesteps, sref.inference_timesteps(20, 1000, "leading", 1))
    out["timesteps.leading.20.offset1"] = ref.timesteps
    try:
        RefScheduler(device=CPU).set_timesteps(1001)
        raise AssertionError("reference did not raise")
    except ValueError:
        pass

    # step / add_noise with injected noise on fixed (4,4,16,16) inputs
    x_t = seeded((4, 4, 16, 16), 101)
    eps = seeded((4, 4, 16, 16), 102)
    noise = seeded((4, 4, 16, 16), 103)
    out["x_t"], out["eps"], out["noise"] = x_t, eps, noise
    real_randn_like = torch.randn_like
    for sched in ("linear", "cosine"):
        ref = RefScheduler(beta_schedule=sched, device=CPU)
        ref.set_timesteps(50)
        tabs = sref.build_tables(sched, 1000)
        ts = sref.inference_timesteps(50)
        for t in (980, 500, 20, 0):
            torch.randn_like = lambda m: noise.to(m.dtype)
            try:
                o = ref.step(eps, torch.tensor(t), x_t)
            finally:
                torch.randn_like = real_randn_like
            s = sref.previous_timestep(ts, t, True)
            assert s == int(ref.previous_timestep(torch.tensor(t)))
            prev, x0 = sref.step(tabs, eps, t, s, x_t, noise)
            assert torch.equal(prev, o.prev_sample) and torch.equal(x0, o.pred_original_sample), (sched, t)
            out[f"step.{sched}.{t}.prev"] = o.prev_sample
            out[f"step.{sched}.{t}.x0"] = o.pred_original_sample
            out[f"step.{sched}.{t}.s"] = np.int64(s)
    # no set_timesteps -> previous = t-1
    ref = RefScheduler(device=CPU)
    torch.randn_like = lambda m: noise.to(m.dtype)
    try:
        o = ref.step(eps, torch.tensor(777), x_t)
    finally:
        torch.randn_like = real_randn_like
    prev, x0 = sref.step(sref.build_tables(), eps, 777, 776, x_t, noise)
    assert torch.equal(prev, o.prev_sample)
    out["step.linear.777.train.prev"] = o.prev_sample
    out["step.linear.777.train.x0"] = o.pred_original_sample
    # bf16 inputs promote to fp32
    torch.randn_like = lambda m: noise.to(m.dtype)
    try:
        ref.set_timesteps(50)
        o = ref.step(eps.bfloat16(), torch.tensor(500), x_t.bfloat16())
    finally:
        torch.randn_like = real_randn_like
    assert o.prev_sample.dtype == torch.float32
    out["step.linear.500.bf16in.prev"] = o.prev_sample
    out["step.linear.500.bf16in.x0"] = o.pred_original_sample

    tt = torch.tensor([0, 999, 500, 123], dtype=torch.long)
    ref = RefScheduler(device=CPU)
    an = ref.add_noise(x_t, noise, tt)
    assert torch.equal(an, sref.add_noise(sref.build_tables(), x_t, noise, tt))
    out["add_noise.t"] = tt
    out["add_noise.out"] = an
    save("scheduler.npz", **out)
    return meta


# ------------------------------------------------------------------------------- UNet
SMALL = dict(block_out_channels=[32, 64, 64, 64])


def ref_unet(cfg_kwargs, seed, gain):
    net = R.UNet(**cfg_kwargs)
    cfg = uref.UNetConfig(**cfg_kwargs)
    shapes = uref.param_shapes(cfg)
    rsd = net.state_dict()
    assert list(rsd.keys()) == list(shapes.keys()), "state_dict key order mismatch"
    for k, v in rsd.items():
        assert tuple(v.shape) == tuple(shapes[k]), (k, v.shape, shapes[k])
    sd = uref.fill_state_dict(shapes, seed=seed, gain=gain)
    net.load_state_dict(sd)
    return net.eval(), cfg, sd


def gen_unet_small():
    print("unet small (config 1)")
    out = {}
    for tag, gain in (("cond", 1.0), ("dflt", 1.0 / math.sqrt(3.0))):
        net, cfg, sd = ref_unet(SMALL, seed=7, gain=gain)
        x = seeded((4, 4, 32, 32), 11)
        ctx = seeded((4, 77, 768), 12)
        t = torch.tensor([999, 500, 20, 0], dtype=torch.long)
        with torch.no_grad():
            y = net(x, t, ctx)
            taps = {}
            y2 = uref.unet_forward(sd, cfg, x, t, ctx, taps)
        e = rel_l2(y2, y)
        sd64 = {k: v.double() for k, v in sd.items()}
        with torch.no_grad():
            y64 = uref.unet_forward(sd64, cfg, x.double(), t, ctx.double())
        e_ref, e_me = rel_l2(y, y64), rel_l2(y2, y64)
        print(f"  [{tag}] oracle vs reference rel-L2 = {e:.2e} (fp32 noise floor vs fp64 oracle: "
              f"reference {e_ref:.2e}, oracle {e_me:.2e}), out std {float(y.std()):.3f}")
        # whole-net fp32 results sit on a ~6e-5 rounding-noise floor (each block amplifies ~1.5x);
        # the fp64 run of the oracle is the tie-breaker: both must be equally close to it.
        assert e < 2e-4 and e_me < 2e-4 and e_ref < 2e-4
        out[f"{tag}.out_f64"] = y64
        out[f"{tag}.out"] = y
        out[f"{tag}.x_sha"] = sha(x)
        out[f"{tag}.ctx_sha"] = sha(ctx)
        for name in ("temb", "mid_block", "down_blocks.3"):
            out[f"{tag}.tap.{name}"] = taps[name]
        for name in ("conv_in", "down_blocks.0", "up_blocks.2"):  # first sample only (fixture size)
            out[f"{tag}.tap0.{name}"] = taps[name][:1]
        # scalar / 0-d timestep forms (unet.py:161-170)
        with torch.no_grad():
            out[f"{tag}.out_int_t"] = net(x, 37, ctx)
            out[f"{tag}.out_float_t"] = net(x, 37.5, ctx)
            out[f"{tag}.out_0d_t"] = net(x, torch.tensor(980), ctx)
            assert rel_l2(uref.unet_forward(sd, cfg, x, 37.5, ctx), out[f"{tag}.out_float_t"]) < 2e-4
    # training-step golden (C2): loss + per-parameter grad norms on the conditioned filler
    net, cfg, sd = ref_unet(SMALL, seed=7, gain=1.0)
    net.train()
    lat = seeded((4, 4, 32, 32), 21)
    noise = seeded((4, 4, 32, 32), 22)
    ctx = seeded((4, 77, 768), 12)
    t = torch.tensor([3, 250, 640, 999], dtype=torch.long)
    sch = RefScheduler(device=CPU)
    noisy = sch.add_noise(lat, noise, t)
    pred = net(noisy, t, ctx)
    loss = torch.nn.functional.mse_loss(pred.float(), noise.float(), reduction="mean")
    loss.backward()
    out["train.loss"] = loss.detach()
    out["train.grad_norms"] = torch.stack([p.grad.norm() for p in net.parameters()])
    out["train.grad.conv_in.weight"] = net.conv_in.weight.grad
    out["train.grad.final_conv.2.weight"] = net.final_conv[2].weight.grad
    out["train.pred"] = pred.detach()
    save("unet_small.npz", **out)


def gen_trajectory():
    """Config 1: 50-step DDPM sampling of 4 latents 4x32x32, small UNet, fp32, injected per-step noise."""
    print("trajectory (config 1)")
    net, cfg, sd = ref_unet(SMALL, seed=7, gain=1.0)
    sch = RefScheduler(device=CPU)
    sch.set_timesteps(50, device="cpu")
    lat = seeded((4, 4, 32, 32), 31) * sch.init_noise_sigma
    ctx = seeded((4, 77, 768), 12)
    out = {"timesteps": sch.timesteps.clone()}
    real_randn_like = torch.randn_like
    keep = (0, 1, 24, 49)
    with torch.no_grad():
        for i, t in enumerate(sch.timesteps):
            eps = net(sch.scale_model_input(lat, t), t, ctx)
            nz = seeded((4, 4, 32, 32), 1000 + i)
            torch.randn_like = lambda m: nz.to(m.dtype)
            try:
                lat = sch.step(eps, t, lat).prev_sample
            finally:
                torch.randn_like = real_randn_like
            if i in keep:
                out[f"eps.{i}"] = eps
                out[f"lat.{i}"] = lat
            if i + 1 in keep:
                out[f"lat.{i}"] = lat  # teacher-forcing input of step i+1
    out["final"] = lat
    print(f"  final latents std {float(lat.std()):.3f}")
    save("trajectory_small.npz", **out)


def gen_blocks():
    """Per-block goldens at the default model's widths (small spatial extents)."""
    print("blocks at default widths")
    out = {}
    T = 1280

    def fill(mod, prefix, seed):
        shapes = {f"{prefix}.{k}": tuple(v.shape) for k, v in mod.state_dict().items()}
        sd = uref.fill_state_dict(shapes, seed=seed, gain=1.0)
        mod.load_state_dict({k[len(prefix) + 1:]: v for k, v in sd.items()})
        return sd

    # ResnetBlock with 1x1 shortcut (320 -> 640) and without (1280 -> 1280)
    for name, ci, co, hw in (("res_sc", 320, 640, 8), ("res_id", 1280, 1280, 4), ("res_cat", 960, 640, 8)):
        m = R.ResnetBlock(ci, co, temb_channels=T).eval()
        sd = fill(m, name, 40)
        x = seeded((1, ci, hw, hw), 41)
        temb = seeded((1, T), 42)
        with torch.no_grad():
            y = m(x, temb)
            y2 = uref.resnet_block(sd, name, x, temb, 32)
        assert rel_l2(y2, y) < 2e-6, name
        out[name + ".out"] = y
    # Attention: self d=40/80/160, cross with 77 tokens
    for name, C, L, cross in (("attn_d40", 320, 256, False), ("attn_d80", 640, 128, False),
                              ("attn_d160", 1280, 64, False), ("xattn_d40", 320, 256, True),
                              ("xattn_d160", 1280, 64, True)):
        m = R.Attention(C, num_heads=8, dim_head=C // 8, cross_attention_dim=768 if cross else None).eval()
        sd = fill(m, name, 50)
        x = seeded((1, L, C), 51)
        ctx = seeded((1, 77, 768), 52) if cross else None
        with torch.no_grad():
            y = m(x, ctx)
            y2 = uref.attention(sd, name, x, ctx if cross else x, 8)
        assert rel_l2(y2, y) < 2e-6, name
        out[name + ".out"] = y
    # TransformerBlock / TransformerModel at C=320
    m = R.TransformerBlock(320, 8, 768).eval()
    sd = fill(m, "tblock", 60)
    x = seeded((1, 64, 320), 61)
    ctx = seeded((1, 77, 768), 62)
    with torch.no_grad():
        y = m(x, ctx)
        assert rel_l2(uref.transformer_block(sd, "tblock", x, ctx, 8, "geglu"), y) < 2e-6
    out["tblock.out"] = y
    m = R.TransformerBlock(320, 8, 768, activation_fn="swiglu").eval()
    sd = fill(m, "tblock_swiglu", 63)
    with torch.no_grad():
        y = m(x, ctx)
        assert rel_l2(uref.transformer_block(sd, "tblock_swiglu", x, ctx, 8, "swiglu"), y) < 2e-6
    out["tblock_swiglu.out"] = y
    m = R.TransformerModel(640, 8, 1, 768).eval()
    sd = fill(m, "tmodel", 70)
    x = seeded((1, 640, 8, 8), 71)
    with torch.no_grad():
        y = m(x, ctx)
        assert rel_l2(uref.transformer_model(sd, "tmodel", x, ctx, uref.UNetConfig()), y) < 2e-6
    out["tmodel.out"] = y
    save("blocks.npz", **out)


def gen_default_meta(meta):
    """Facts about the default (SD-1.5-shaped) UNet that need the reference to establish."""
    net = R.UNet()
    cfg = uref.UNetConfig()
    shapes = uref.param_shapes(cfg)
    rsd = net.state_dict()
    assert list(rsd.keys()) == list(shapes.keys())
    assert all(tuple(rsd[k].shape) == tuple(shapes[k]) for k in shapes)
    meta["default.num_params"] = int(sum(v.numel() for v in rsd.values()))
    meta["default.num_tensors"] = len(rsd)
    meta["default.flops_64"] = uref.flops_per_image(cfg, 64, 64)
    meta["default.flops_32"] = uref.flops_per_image(cfg, 32, 32)
    meta["default.flops_96"] = uref.flops_per_image(cfg, 96, 96)
    print("  default UNet:", meta["default.num_params"], "params;",
          f"{meta['default.flops_64'] / 1e9:.2f} GFLOP/img @64x64")
    # one default-width forward at 16x16 (B=1) as a whole-model golden for the big config
    sd = uref.fill_state_dict(shapes, seed=3, gain=1.0)
    net.load_state_dict(sd)
    net.eval()
    x = seeded((1, 4, 16, 16), 81)
    ctx = seeded((1, 77, 768), 82)
    with torch.no_grad():
        y = net(x, torch.tensor([431]), ctx)
        y2 = uref.unet_forward(sd, cfg, x, torch.tensor([431]), ctx)
    e = rel_l2(y2, y)
    sd64 = {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        y64 = uref.unet_forward(sd64, cfg, x.double(), torch.tensor([431]), ctx.double())
    print(f"  default 16x16 oracle vs reference rel-L2 {e:.2e}; vs fp64 oracle: reference "
          f"{rel_l2(y, y64):.2e}, oracle {rel_l2(y2, y64):.2e}")
    assert e < 2e-4  # fp32 rounding-noise floor of the whole net, see gen_unet_small
    save("unet_default16.npz", out=y, out_f64=y64)


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    meta = gen_scheduler()
    gen_blocks()
    gen_unet_small()
    gen_trajectory()
    gen_default_meta(meta)
    meta["torch"] = torch.__version__
    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("done")
