"""CPU: the oracle restatement against the golden vectors generated from the reference
(tests/golden/make_golden.py).  These pin the checker the GPU parity tests rely on."""
import hashlib
import json
import os

import pytest
import torch

from conftest import GOLDEN, rel_l2, same_bits, seeded
from oracle import scheduler_ref as sref
from oracle import unet_ref as uref

SMALL = dict(block_out_channels=[32, 64, 64, 64])


def _sha(t):
    return hashlib.sha256(t.contiguous().numpy().tobytes()).hexdigest()[:16]


@pytest.mark.parametrize("sched", ["linear", "cosine", "sigmoid"])
def test_tables_bit_exact(golden, sched):
    g = golden("scheduler.npz")
    meta = json.load(open(os.path.join(GOLDEN, "meta.json")))
    tabs = sref.build_tables(sched, 1000)
    for k, v in tabs.items():
        assert same_bits(v, g[f"{sched}.{k}"]), k
        assert _sha(v) == meta[f"{sched}.{k}"]


def test_table_kats_from_survey():
    """Known-answer values recorded in SURVEY.md section 8a/S1 (big-endian fp32 hex at idx 0,1,500,999)."""
    t = sref.build_tables("linear", 1000)

    def hx(x, i):
        return x[i].view(torch.int32).item() & 0xffffffff

    assert [hx(t["betas"], i) for i in (0, 1, 500, 999)] == [0x38d1b717, 0x38fb7d84, 0x3c24d288, 0x3ca3d70a]
    assert [hx(t["alphas_cumprod"], i) for i in (0, 1, 500, 999)] == [0x3f7ff972, 0x3f7ff196, 0x3d9f53db, 0x38294665]
    assert _sha(t["alphas_cumprod"]) == "4ae258e7e3ffb43a"
    assert _sha(t["posterior_std"]) == "fd78de2d699b1ebe"


@pytest.mark.parametrize("spacing", ["leading", "linspace", "trailing"])
def test_timesteps(golden, spacing):
    g = golden("scheduler.npz")
    for n in (50, 10, 3, 1000, 7):
        assert same_bits(sref.inference_timesteps(n, 1000, spacing), g[f"timesteps.{spacing}.{n}"])
    assert sref.inference_timesteps(50).tolist()[:3] == [980, 960, 940]
    with pytest.raises(ValueError):
        sref.inference_timesteps(1001)


def test_step_and_add_noise(golden):
    g = golden("scheduler.npz")
    ts = sref.inference_timesteps(50)
    for sched in ("linear", "cosine"):
        tabs = sref.build_tables(sched)
        for t in (980, 500, 20, 0):
            s = sref.previous_timestep(ts, t, True)
            assert s == int(g[f"step.{sched}.{t}.s"])
            prev, x0 = sref.step(tabs, g["eps"], t, s, g["x_t"], g["noise"])
            assert same_bits(prev, g[f"step.{sched}.{t}.prev"])
            assert same_bits(x0, g[f"step.{sched}.{t}.x0"])
    tabs = sref.build_tables()
    prev, _ = sref.step(tabs, g["eps"], 777, 776, g["x_t"], g["noise"])
    assert same_bits(prev, g["step.linear.777.train.prev"])
    prev, _ = sref.step(tabs, g["eps"].bfloat16(), 500, 480, g["x_t"].bfloat16(), g["noise"])
    assert prev.dtype == torch.float32 and same_bits(prev, g["step.linear.500.bf16in.prev"])
    out = sref.add_noise(tabs, g["x_t"], g["noise"], g["add_noise.t"])
    assert same_bits(out, g["add_noise.out"])


def test_state_dict_contract():
    meta = json.load(open(os.path.join(GOLDEN, "meta.json")))
    shapes = uref.param_shapes(uref.UNetConfig())
    assert len(shapes) == meta["default.num_tensors"] == 654
    n = 0
    for s in shapes.values():
        k = 1
        for d in s:
            k *= d
        n += k
    assert n == meta["default.num_params"] == 859520964
    assert shapes["conv_in.weight"] == (320, 4, 3, 3)
    assert shapes["down_blocks.0.attentions.0.transformer_blocks.0.feed_forward.1.w1_w2.weight"] == (2560, 320)
    assert shapes["up_blocks.1.resnets.2.visual.2.weight"] == (1280, 1920, 3, 3)
    assert abs(uref.flops_per_image(uref.UNetConfig(), 64, 64) / 803.27e9 - 1) < 1e-4


def test_blocks(golden):
    g = golden("blocks.npz")
    T = 1280

    def sd_for(prefix, shapes, seed):
        return uref.fill_state_dict({f"{prefix}.{k}": v for k, v in shapes.items()}, seed=seed, gain=1.0)

    def resnet_shapes(ci, co):
        s = {"visual.0.weight": (ci,), "visual.0.bias": (ci,), "visual.2.weight": (co, ci, 3, 3), "visual.2.bias": (co,),
             "time.1.weight": (co, T), "time.1.bias": (co,), "layers.0.weight": (co,), "layers.0.bias": (co,),
             "layers.3.weight": (co, co, 3, 3), "layers.3.bias": (co,)}
        if ci != co:
            s["conv_shortcut.weight"] = (co, ci, 1, 1)
            s["conv_shortcut.bias"] = (co,)
        return s

    for name, ci, co, hw in (("res_sc", 320, 640, 8), ("res_id", 1280, 1280, 4), ("res_cat", 960, 640, 8)):
        sd = sd_for(name, resnet_shapes(ci, co), 40)
        y = uref.resnet_block(sd, name, seeded((1, ci, hw, hw), 41), seeded((1, T), 42), 32)
        assert rel_l2(y, g[name + ".out"]) < 2e-6, name

    def attn_shapes(C, cd):
        return {"q_proj.weight": (C, C), "kv_proj.weight": (2 * C, cd), "output_proj.0.weight": (C, C),
                "output_proj.0.bias": (C,)}

    for name, C, L, cross in (("attn_d40", 320, 256, False), ("attn_d80", 640, 128, False), ("attn_d160", 1280, 64, False),
                              ("xattn_d40", 320, 256, True), ("xattn_d160", 1280, 64, True)):
        sd = sd_for(name, attn_shapes(C, 768 if cross else C), 50)
        x = seeded((1, L, C), 51)
        ctx = seeded((1, 77, 768), 52) if cross else x
        assert rel_l2(uref.attention(sd, name, x, ctx, 8), g[name + ".out"]) < 2e-6, name


def _small(gain):
    cfg = uref.UNetConfig(**SMALL)
    return cfg, uref.fill_state_dict(uref.param_shapes(cfg), seed=7, gain=gain)


def test_small_unet_forward(golden):
    g = golden("unet_small.npz")
    x, ctx = seeded((4, 4, 32, 32), 11), seeded((4, 77, 768), 12)
    assert _sha(x) == g["cond.x_sha"] and _sha(ctx) == g["cond.ctx_sha"], "CPU generator drifted from the fixture's"
    t = torch.tensor([999, 500, 20, 0])
    for tag, gain in (("cond", 1.0), ("dflt", 3 ** -0.5)):
        cfg, sd = _small(gain)
        with torch.no_grad():
            taps = {}
            y = uref.unet_forward(sd, cfg, x, t, ctx, taps)
        # whole-net fp32 runs sit on a ~6e-5 rounding-noise floor (see make_golden.py); 2e-4 bounds it
        assert rel_l2(y, g[f"{tag}.out"]) < 2e-4
        assert rel_l2(y, g[f"{tag}.out_f64"]) < 2e-4
        assert rel_l2(taps["temb"], g[f"{tag}.tap.temb"]) < 1e-6
        assert rel_l2(taps["conv_in"][:1], g[f"{tag}.tap0.conv_in"]) < 1e-6
        assert rel_l2(taps["mid_block"], g[f"{tag}.tap.mid_block"]) < 1e-4
    cfg, sd = _small(1.0)
    with torch.no_grad():
        assert rel_l2(uref.unet_forward(sd, cfg, x, 37, ctx), g["cond.out_int_t"]) < 2e-4
        assert rel_l2(uref.unet_forward(sd, cfg, x, 37.5, ctx), g["cond.out_float_t"]) < 2e-4
        assert rel_l2(uref.unet_forward(sd, cfg, x, torch.tensor(980), ctx), g["cond.out_0d_t"]) < 2e-4


def test_small_unet_train_step(golden):
    """C2: noise-prediction MSE + backward through the oracle reproduces the reference's loss and gradients."""
    g = golden("unet_small.npz")
    cfg, sd = _small(1.0)
    sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    lat, noise, ctx = seeded((4, 4, 32, 32), 21), seeded((4, 4, 32, 32), 22), seeded((4, 77, 768), 12)
    t = torch.tensor([3, 250, 640, 999])
    noisy = sref.add_noise(sref.build_tables(), lat, noise, t)
    pred = uref.unet_forward(sd, cfg, noisy, t, ctx)
    loss = torch.nn.functional.mse_loss(pred.float(), noise.float())
    loss.backward()
    assert abs(float(loss) - float(g["train.loss"])) < 1e-4 * float(g["train.loss"])
    norms = torch.stack([v.grad.norm() for v in sd.values()])
    assert rel_l2(norms, g["train.grad_norms"]) < 1e-3
    assert rel_l2(sd["conv_in.weight"].grad, g["train.grad.conv_in.weight"]) < 1e-3


def test_trajectory_teacher_forced(golden):
    """Config 1 trajectory: with the reference's latents fed in at steps 0/1/24/49 the oracle reproduces the
    per-step noise prediction and the scheduler update."""
    g = golden("trajectory_small.npz")
    cfg, sd = _small(1.0)
    tabs = sref.build_tables()
    ts = g["timesteps"]
    assert same_bits(ts, sref.inference_timesteps(50))
    ctx = seeded((4, 77, 768), 12)
    for i in (0, 1, 24, 49):
        lat_in = seeded((4, 4, 32, 32), 31) if i == 0 else g[f"lat.{i - 1}"]
        t = int(ts[i])
        with torch.no_grad():
            eps = uref.unet_forward(sd, cfg, lat_in, t, ctx)
        assert rel_l2(eps, g[f"eps.{i}"]) < 2e-4
        prev, _ = sref.step(tabs, g[f"eps.{i}"], t, sref.previous_timestep(ts, t, True), lat_in,
                            seeded((4, 4, 32, 32), 1000 + i))
        assert same_bits(prev, g[f"lat.{i}"])


# ------------------------------------------------------------------------- round 2: S5 and the new fixtures
def _s5_inputs(g):
    return g["x_t"], g["x0"], g["noise"], g["t"], g["s"]


@pytest.mark.parametrize("sched", ["linear", "cosine"])
def test_s5_methods_oracle_and_product_host_side(golden, sched):
    """S5 (scheduling_ddpm.py:103-129, :217-256): the oracle restatement and the product's methods (plain torch ops,
    here on CPU tensors) against the reference's outputs, bit for bit; per-sample t, s = -1 on the last sample."""
    from ldm.scheduling_ddpm import DDPMScheduler
    g = golden("scheduler_s5.npz")
    x_t, x0, nz, t, s = _s5_inputs(g)
    tabs = sref.build_tables(sched, 1000)
    sch = DDPMScheduler(beta_schedule=sched, device=torch.device("cpu"))
    for name, fn in (("oracle", None), ("product", sch)):
        if fn is None:
            got = {"start": sref.predict_start_from_noise(tabs, x_t, t, nz), "noise": sref.predict_noise_from_start(tabs, x_t, t, x0)}
            got["qp.mean"], got["qp.std"] = sref.q_posterior(tabs, x0, x_t, t)
            got["qb.mean"], got["qb.std"] = sref.q_posterior_between(tabs, x0, x_t, t, s)
        else:
            got = {"start": fn.predict_start_from_noise(x_t, t, nz), "noise": fn.predict_noise_from_start(x_t, t, x0)}
            got["qp.mean"], got["qp.std"] = fn.q_posterior(x0, x_t, t)
            got["qb.mean"], got["qb.std"] = fn.q_posterior_between(x0, x_t, t, s)
        for k, v in got.items():
            assert same_bits(v, g[f"{sched}.{k}"]), (name, sched, k)
    assert same_bits(sch.scale_model_input(x_t, 5), x_t)


def test_scheduler_module_level_helpers(golden):
    """extract / linear_ / cosine_ / sigmoid_beta_schedule / get_beta_schedule of the drop-in module (:285-357)."""
    import ldm.scheduling_ddpm as S
    g = golden("scheduler_s5.npz")
    for sched, fn in (("linear", S.linear_beta_schedule), ("cosine", S.cosine_beta_schedule), ("sigmoid", S.sigmoid_beta_schedule)):
        for T in (1000, 250):
            b = S.get_beta_schedule(sched, T)
            assert b.dtype == torch.float64 and same_bits(b, g[f"betas64.{sched}.{T}"]) and same_bits(fn(T), b)
            assert same_bits(sref.betas_f64(sched, T), b)
    with pytest.raises(ValueError):
        S.get_beta_schedule("bogus", 10)
    a = torch.arange(10, dtype=torch.float32)
    e = S.extract(a, torch.tensor([3, 0, 9]), (3, 4, 8, 8))
    assert e.shape == (3, 1, 1, 1) and e.flatten().tolist() == [3.0, 0.0, 9.0]


def test_well_conditioned_filler_small(golden):
    """unet_wellcond.npz: the oracle on the well-conditioned filler reproduces the reference's float32 output, and that
    filler leaves every tensor but the gate halves of the GLU projections untouched."""
    g = golden("unet_wellcond.npz")
    cfg = uref.UNetConfig(**SMALL)
    shapes = uref.param_shapes(cfg)
    sd = uref.fill_state_dict(shapes, seed=7, **uref.WELL_CONDITIONED)
    plain = uref.fill_state_dict(shapes, seed=7, gain=1.0)
    changed = [k for k in sd if not torch.equal(sd[k], plain[k])]
    assert changed and all(".w1_w2." in k for k in changed)
    k = changed[0]
    half = sd[k].shape[0] // 2
    assert torch.equal(sd[k][:half], plain[k][:half])
    x, ctx = seeded((4, 4, 32, 32), 11), seeded((4, 77, 768), 12)
    with torch.no_grad():
        y = uref.unet_forward(sd, cfg, x, torch.tensor([999, 500, 20, 0]), ctx)
    assert rel_l2(y, g["small.out"]) < 1e-5 and rel_l2(y, g["small.out_f64"]) < 1e-5
    assert float(g["small.err_torch_bf16"]) < 4e-2 and float(g["traj.psnr_torch_bf16"]) > 40.0
