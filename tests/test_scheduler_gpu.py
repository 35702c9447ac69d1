"""GPU parity: the fused HIP scheduler kernels (through the C-ABI) against the CPU oracle and the golden
vectors taken from the reference.

Tolerance: BIT-EXACT against the oracle evaluated on the same host (float32 IEEE operations in the reference's
order).  Against the golden files (generated in the build container) the tables that go through torch's CPU
sqrt are allowed 1 ulp (4 ulp where a product and a quotient follow the square root) in at most 320 of 1000 entries (measured on the boxes of the pool: 6 ... 285; the counts of the last run are kept in profiles/) and what is computed from them 1e-6 relative L2: torch's CPU sqrt is MKL's, which is not correctly
rounded and differs between host CPUs (measured: 6/1000 entries of sqrt_alphas_cumprod differ by 1 ulp between
the build container and the GPU box), so the reference itself is only reproducible to that level across hosts.
betas, alphas_cumprod and the timestep lists are bit-exact against the golden files everywhere."""
import pytest
import torch

from conftest import max_ulp, rel_l2, same_bits, seeded
from oracle import scheduler_ref as sref

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sched():
    from ldm.scheduling_ddpm import DDPMScheduler
    return DDPMScheduler(device=torch.device("cuda"))


def test_tables_on_device_bit_exact(golden, sched):
    g = golden("scheduler.npz")
    host = sref.build_tables("linear", 1000)
    for k in ("betas", "alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
              "posterior_mean_coef1", "posterior_mean_coef2", "posterior_std"):
        assert getattr(sched, k).is_cuda
        assert same_bits(getattr(sched, k), host[k]), k
        if k in ("betas", "alphas_cumprod"):
            assert same_bits(getattr(sched, k), g[f"linear.{k}"]), k
        else:
            # torch's CPU sqrt (MKL) on this host against the build container's: never more than 1 ulp, in a minority of the
            # entries (measured on boxes of the pool: 6 and 187 of 1000 for sqrt_alphas_cumprod -- it depends on the host CPU)
            dev, ref = getattr(sched, k).cpu(), g[f"linear.{k}"]
            ndiff = int((dev != ref).sum())
            print(f"{k}: {ndiff}/1000 entries differ from the build container's by 1 ulp")
            # a table that IS a square root: 1 ulp; posterior_mean_coef1/2 multiply and divide a 1-ulp-off square root: 4
            assert max_ulp(dev, ref) <= (4 if k.startswith("posterior_mean") else 1), k
            assert ndiff <= 320, (k, ndiff)      # measured over the pool's hosts: 6 ... 285


@pytest.mark.parametrize("schedule", ["linear", "cosine"])
def test_step_golden_bit_exact(golden, schedule):
    from ldm.scheduling_ddpm import DDPMScheduler
    g = golden("scheduler.npz")
    s = DDPMScheduler(beta_schedule=schedule, device=torch.device("cuda"))
    s.set_timesteps(50, device="cuda")
    x, e, z = g["x_t"].cuda(), g["eps"].cuda(), g["noise"].cuda()
    tabs = sref.build_tables(schedule)
    ts = sref.inference_timesteps(50)
    for t in (980, 500, 20, 0):
        out = s.step(e, torch.tensor(t), x, noise=z)
        assert out.prev_sample.dtype == torch.float32
        ref_prev, ref_x0 = sref.step(tabs, g["eps"], t, sref.previous_timestep(ts, t, True), g["x_t"], g["noise"])
        assert same_bits(out.prev_sample, ref_prev), t
        assert same_bits(out.pred_original_sample, ref_x0), t
        assert rel_l2(out.prev_sample, g[f"step.{schedule}.{t}.prev"]) < 1e-6, t
        assert rel_l2(out.pred_original_sample, g[f"step.{schedule}.{t}.x0"]) < 1e-6, t


def test_step_iterating_device_timesteps_no_sync(golden, sched):
    """`for t in scheduler.timesteps` hands step() 0-d device tensors (train_latent_diffusion.py:579)."""
    g = golden("scheduler.npz")
    sched.set_timesteps(50)
    x, e, z = g["x_t"].cuda(), g["eps"].cuda(), g["noise"].cuda()
    seen = []
    for t in sched.timesteps:
        seen.append(sched._timestep_value(t))
        if seen[-1] == 500:
            out = sched.step(e, t, x, noise=z)
            ref, _ = sref.step(sref.build_tables(), g["eps"], 500, 480, g["x_t"], g["noise"])
            assert same_bits(out.prev_sample, ref)
            assert rel_l2(out.prev_sample, g["step.linear.500.prev"]) < 1e-5
    assert seen == sref.inference_timesteps(50).tolist()
    sched.num_inference_steps = None  # training-style: previous = t - 1
    out = sched.step(e, 777, x, noise=z)
    ref, _ = sref.step(sref.build_tables(), g["eps"], 777, 776, g["x_t"], g["noise"])
    assert same_bits(out.prev_sample, ref)
    assert rel_l2(out.prev_sample, g["step.linear.777.train.prev"]) < 1e-5


def test_step_bf16_inputs_promote(golden, sched):
    g = golden("scheduler.npz")
    sched.set_timesteps(50)
    out = sched.step(g["eps"].bfloat16().cuda(), 500, g["x_t"].bfloat16().cuda(), noise=g["noise"].cuda())
    assert out.prev_sample.dtype == torch.float32
    ref, ref0 = sref.step(sref.build_tables(), g["eps"].bfloat16(), 500, 480, g["x_t"].bfloat16(), g["noise"])
    assert same_bits(out.prev_sample, ref) and same_bits(out.pred_original_sample, ref0)
    assert rel_l2(out.prev_sample, g["step.linear.500.bf16in.prev"]) < 1e-5
    # mixed: bf16 eps from the UNet, float32 latents (what generate() has after the first step)
    out = sched.step(g["eps"].bfloat16().cuda(), 500, g["x_t"].cuda(), noise=g["noise"].cuda())
    ref, _ = sref.step(sref.build_tables(), g["eps"].bfloat16(), 500, 480, g["x_t"], g["noise"])
    assert same_bits(out.prev_sample, ref)


@pytest.mark.parametrize("shape", [(1, 4, 8, 8), (3, 4, 5, 7), (2, 3, 1, 1), (1, 1, 1, 1), (0, 4, 8, 8)])
def test_step_ragged_sizes(sched, shape):
    sched.set_timesteps(10)
    tabs = sref.build_tables()
    x, e, z = seeded(shape, 1), seeded(shape, 2), seeded(shape, 3)
    for t, s in ((900, 800), (0, -1)):
        out = sched.step(e.cuda(), t, x.cuda(), noise=z.cuda())
        ref_prev, ref_x0 = sref.step(tabs, e, t, s, x, z)
        assert out.prev_sample.shape == x.shape
        assert same_bits(out.prev_sample, ref_prev)
        assert same_bits(out.pred_original_sample, ref_x0)


def test_step_cfg_fused(sched):
    sched.set_timesteps(50)
    shape = (4, 4, 32, 32)
    x, eu, ec, z = seeded(shape, 1), seeded(shape, 2), seeded(shape, 3), seeded(shape, 4)
    eps = eu + 7.5 * (ec - eu)  # train_latent_diffusion.py:592
    ref, _ = sref.step(sref.build_tables(), eps, 500, 480, x, z)
    out = sched.step(eu.cuda(), 500, x.cuda(), noise=z.cuda(), guidance=(ec.cuda(), 7.5))
    assert same_bits(out.prev_sample, ref)


def test_step_full_size_and_default_noise(sched):
    """cfg3 shape (256,4,64,64): bit-exact against the oracle, and the default draw comes from the global
    device generator (scheduling_ddpm.py:212)."""
    sched.set_timesteps(50)
    shape = (256, 4, 64, 64)
    x, e, z = seeded(shape, 5), seeded(shape, 6), seeded(shape, 7)
    ref, ref0 = sref.step(sref.build_tables(), e, 980, 960, x, z)
    out = sched.step(e.cuda(), 980, x.cuda(), noise=z.cuda())
    assert same_bits(out.prev_sample, ref) and same_bits(out.pred_original_sample, ref0)
    torch.manual_seed(123)
    a = sched.step(e.cuda(), 980, x.cuda()).prev_sample
    torch.manual_seed(123)
    zz = torch.randn(shape, dtype=torch.float32, device="cuda")
    b = sched.step(e.cuda(), 980, x.cuda(), noise=zz).prev_sample
    assert same_bits(a, b)
    # last step: std floor sqrt(1e-20) = 1e-10 and alpha_bar_{-1} := 1 -> prev == x0 up to 1e-10*noise
    last = sched.step(e.cuda(), 0, x.cuda(), noise=z.cuda())
    assert torch.allclose(last.prev_sample, last.pred_original_sample, atol=1e-6, rtol=0)


def test_add_noise(golden, sched):
    g = golden("scheduler.npz")
    out = sched.add_noise(g["x_t"].cuda(), g["noise"].cuda(), g["add_noise.t"].cuda())
    tabs = sref.build_tables()
    assert same_bits(out, sref.add_noise(tabs, g["x_t"], g["noise"], g["add_noise.t"]))
    assert rel_l2(out, g["add_noise.out"]) < 1e-5
    for shape in ((32, 4, 64, 64), (3, 4, 5, 7), (2, 1, 1, 1)):
        x, z = seeded(shape, 8), seeded(shape, 9)
        t = torch.randint(0, 1000, (shape[0],), generator=torch.Generator().manual_seed(10))
        out = sched.add_noise(x.cuda(), z.cuda(), t.cuda())
        assert same_bits(out, sref.add_noise(tabs, x, z, t)), shape
        outb = sched.add_noise(x.bfloat16().cuda(), z.bfloat16().cuda(), t.cuda())
        refb = sref.add_noise(tabs, x.bfloat16(), z.bfloat16(), t)
        assert outb.dtype == refb.dtype == torch.float32 and same_bits(outb, refb), shape
    with pytest.raises(ValueError):
        sched.add_noise(g["x_t"].cuda(), g["noise"].cuda(), torch.zeros(3, dtype=torch.long, device="cuda"))
    # a timestep outside [0, T) (the reference's gather asserts on the device): that sample comes back as NaN, the others intact
    tbad = g["add_noise.t"].clone()
    tbad[1], tbad[3] = 1000, -1
    outb = sched.add_noise(g["x_t"].cuda(), g["noise"].cuda(), tbad.cuda())
    assert torch.isnan(outb[1]).all() and torch.isnan(outb[3]).all()
    good = sref.add_noise(tabs, g["x_t"], g["noise"], g["add_noise.t"])
    assert same_bits(outb[0], good[0]) and same_bits(outb[2], good[2])


def test_fused_training_input_prep_matches_the_reference_sequence():
    """ldm/train_inputs.py against the torch op sequence of train_latent_diffusion.py:1039-1079 with the same
    generator: posterior sample x scaling, offset noise, input perturbation, add_noise (oracle), conditioning dropout."""
    import torch
    from ldm import train_inputs
    from ldm.scheduling_ddpm import DDPMScheduler
    from oracle import scheduler_ref as sref
    dev = torch.device("cuda")
    sch = DDPMScheduler(device=dev)
    moments = torch.randn(5, 8, 16, 12, generator=torch.Generator().manual_seed(3)).cuda()
    moments[:, 4:] = moments[:, 4:] * 20      # logvar beyond the clamp on both sides
    for off, pert in ((0.0, 0.0), (0.1, 0.0), (0.05, 0.2)):
        got = train_inputs.prepare_inputs(sch, moments, 0.18215, noise_offset=off, input_perturbation=pert,
                                          generator=torch.Generator(device=dev).manual_seed(9))
        g = torch.Generator(device=dev).manual_seed(9)
        mean, logvar = moments.float().chunk(2, dim=1)
        std = torch.exp(0.5 * torch.clamp(logvar, -30.0, 20.0))
        lat = 0.18215 * (mean + std * torch.randn(mean.shape, device=dev, generator=g))
        noise = torch.randn(lat.shape, device=dev, generator=g)
        if off:
            noise = noise + off * torch.randn((5, 4, 1, 1), device=dev, generator=g)
        nin = noise + pert * torch.randn(lat.shape, device=dev, generator=g) if pert else noise
        t = torch.randint(0, 1000, (5,), dtype=torch.int64, device=dev, generator=g)
        noisy = sref.add_noise(sref.build_tables(), lat.cpu(), nin.cpu(), t.cpu())
        assert torch.equal(got.timesteps, t)
        assert rel_l2(got.latents, lat) < 1e-6 and rel_l2(got.noise, noise) < 1e-7
        assert rel_l2(got.noisy_latents, noisy) < 1e-6
    enc = torch.randn(6, 77, 768, device=dev).bfloat16()
    null = torch.randn(1, 77, 768, device=dev).bfloat16()
    mask = torch.tensor([0, 1, 0, 0, 1, 1], dtype=torch.bool, device=dev)
    ref = enc.clone()
    ref[mask] = null
    assert torch.equal(train_inputs.drop_conditioning(enc, null, mask), ref)


@pytest.mark.parametrize("schedule", ["linear", "cosine"])
def test_s5_methods_on_device(golden, schedule):
    """S5 (scheduling_ddpm.py:103-129, :217-256) on GPU tensors: table gathers and float32 elementwise arithmetic, so
    bit-exact against the reference wherever no device sqrt is involved (predict_*, q_posterior); q_posterior_between takes
    two square roots on the device -- IEEE-exact there, MKL's on the reference's CPU -- hence 1e-6."""
    from ldm.scheduling_ddpm import DDPMScheduler
    g = golden("scheduler_s5.npz")
    sch = DDPMScheduler(beta_schedule=schedule, device=torch.device("cuda"))
    x_t, x0, nz, t, s = (g[k].cuda() for k in ("x_t", "x0", "noise", "t", "s"))
    tabs = sref.build_tables(schedule)
    a = sch.predict_start_from_noise(x_t, t, nz)
    b = sch.predict_noise_from_start(x_t, t, x0)
    assert same_bits(a, sref.predict_start_from_noise(tabs, g["x_t"], g["t"], g["noise"]))
    assert same_bits(b, sref.predict_noise_from_start(tabs, g["x_t"], g["t"], g["x0"]))
    assert rel_l2(a, g[f"{schedule}.start"]) < 1e-6 and rel_l2(b, g[f"{schedule}.noise"]) < 1e-6
    m, sd = sch.q_posterior(x0, x_t, t)
    rm, rs = sref.q_posterior(tabs, g["x0"], g["x_t"], g["t"])
    assert same_bits(m, rm) and same_bits(sd, rs)
    assert rel_l2(m, g[f"{schedule}.qp.mean"]) < 1e-6 and rel_l2(sd, g[f"{schedule}.qp.std"]) < 1e-6
    m, sd = sch.q_posterior_between(x0, x_t, t, s)
    assert m.shape == x_t.shape and sd.shape == (5, 1, 1, 1)
    assert rel_l2(m, g[f"{schedule}.qb.mean"]) < 1e-6 and rel_l2(sd, g[f"{schedule}.qb.std"]) < 1e-6
    # the last sample has s = -1: alpha_bar_{-1} := 1, std floor sqrt(1e-20)
    assert abs(float(sd[4]) - 1e-10) < 1e-16
    # step() == predict_start_from_noise + q_posterior_between + noise (how the reference composes it, :207-212)
    sch.set_timesteps(50)
    out = sch.step(nz, 500, x_t, noise=x0)
    tt, ss = torch.full((5,), 500, device="cuda"), torch.full((5,), 480, device="cuda")
    x0p = sch.predict_start_from_noise(x_t, tt, nz)
    mean, std = sch.q_posterior_between(x0p, x_t, tt, ss)
    assert rel_l2(out.prev_sample, mean + std * x0) < 1e-6


def test_gaussian_noise_kernel_matches_the_oracle_and_ignores_the_sharding():
    """ldm_gaussian_noise (Philox4x32-10 + Box-Muller keyed by the global sample index): integer part bit-exact by construction,
    the float32 transcendental functions of device and host agree to 2e-6 absolute; rows drawn by a shard are BIT-equal to the
    same rows of the whole batch (what makes the sharded sampler independent of the number of GPUs)."""
    import numpy as np
    from ldm import ops, sampling
    from oracle import noise_ref
    dev = torch.device("cuda")
    seed, step, shape = 0x1234_5678_9ABC_DEF0, 17, (4, 64, 64)
    full = ops.gaussian_noise(seed, step, 0, 32, shape, dev)
    ref = torch.from_numpy(noise_ref.gaussian_rows(seed, step, 0, 32, 4 * 64 * 64)).reshape(32, *shape)
    assert float((full.cpu() - ref).abs().max()) < 2e-6
    part = ops.gaussian_noise(seed, step, 11, 5, shape, dev)
    assert same_bits(part, full[11:16])
    assert same_bits(sampling.global_noise(seed, step, (32, *shape), 11, 16, dev), part)
    assert not torch.equal(ops.gaussian_noise(seed, step + 1, 11, 5, shape, dev), part)
    z = ops.gaussian_noise(3, -1, 1000, 3, (7,), dev)          # ragged rows, initial-latent step
    assert float((z.cpu() - torch.from_numpy(noise_ref.gaussian_rows(3, -1, 1000, 3, 7))).abs().max()) < 2e-6
    big = ops.gaussian_noise(9, 0, 0, 256, shape, dev)
    assert abs(float(big.mean())) < 2e-3 and abs(float(big.std()) - 1.0) < 2e-3 and torch.isfinite(big).all()
    assert ops.gaussian_noise(9, 0, 5, 0, shape, dev).shape == (0, *shape)
    with pytest.raises(RuntimeError):
        ops.gaussian_noise(9, 0, 0, 1, shape, torch.device("cpu"))
