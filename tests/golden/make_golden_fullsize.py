"""Golden vectors at the BENCHMARKED shapes and on the well-conditioned filler (round 2), generated in the build
container by importing the REFERENCE (same two inert stand-ins as make_golden.py, which this script imports).

  unet_default64.npz   default (859.5 M parameter) UNet, B=2, 4x64x64, per-sample timesteps -- BASELINE.json configs[1]'s
                       shape: reference float32 output, float64 oracle output, for the plain (`p.*`) and the well-conditioned
                       (`w.*`) filler, and torch's own bf16 end-to-end error on each
  unet_default96.npz   same net, B=1, 4x96x96 (configs[4]'s extent: 9,216-token self-attention), well-conditioned filler
  unet_wellcond.npz    small (config 1) UNet and default UNet at 16x16 on the well-conditioned filler; a 50-step config-1
                       trajectory on it with the PSNR of torch's bf16 run (calibrates the bf16 PSNR floor)

Every vector is checked against the oracle restatement while it is generated (abort on mismatch).  Data only.
  scheduler_s5.npz     the scheduler methods `step` does not call (S5) and the float64 beta schedules

  unet_default64_moderate.npz   configs[1]'s shape and the small net on the moderate-gate filler (round 3)

Run: python tests/golden/make_golden_fullsize.py [s5] [64] [64m] [96] [wellcond]
"""
from __future__ import annotations

import math
import os
import sys
import time

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402  (installs the stand-ins, imports the reference as mg.R)
from make_golden import rel_l2, save, seeded  # noqa: E402

uref = mg.uref
WELL = uref.WELL_CONDITIONED


def ref_unet(cfg_kwargs, seed, **fill):
    net = mg.R.UNet(**cfg_kwargs)
    cfg = uref.UNetConfig(**cfg_kwargs)
    shapes = uref.param_shapes(cfg)
    sd = uref.fill_state_dict(shapes, seed=seed, **fill)
    net.load_state_dict(sd)
    return net.eval(), cfg, sd


def bf16_oracle(sd, cfg, x, t, ctx):
    with torch.no_grad():
        return uref.unet_forward({k: v.bfloat16() for k, v in sd.items()}, cfg, x.bfloat16(), t, ctx.bfloat16()).float()


def f64_oracle(sd, cfg, x, t, ctx):
    with torch.no_grad():
        return uref.unet_forward({k: v.double() for k, v in sd.items()}, cfg, x.double(), t, ctx.double())


def one_case(tag, cfg_kwargs, seed, fill, x, t, ctx, out, want_bf16=True):
    t0 = time.time()
    net, cfg, sd = ref_unet(cfg_kwargs, seed, **fill)
    with torch.no_grad():
        y = net(x, t, ctx)
        y2 = uref.unet_forward(sd, cfg, x, t, ctx)
    del net
    y64 = f64_oracle(sd, cfg, x, t, ctx)
    e, e_ref, e_me = rel_l2(y2, y), rel_l2(y, y64), rel_l2(y2, y64)
    print(f"  [{tag}] oracle vs reference {e:.2e}; vs f64 oracle: reference {e_ref:.2e}, oracle {e_me:.2e}; "
          f"out std {float(y64.std()):.3f}  ({time.time() - t0:.0f} s)", flush=True)
    assert e < 2e-4 and e_ref < 2e-4 and e_me < 2e-4
    out[f"{tag}.out"] = y
    out[f"{tag}.out_f64"] = y64
    if want_bf16:
        t0 = time.time()
        eb = rel_l2(bf16_oracle(sd, cfg, x, t, ctx), y64)
        print(f"  [{tag}] torch-CPU bf16 end-to-end vs f64 oracle {eb:.3e}  ({time.time() - t0:.0f} s)", flush=True)
        out[f"{tag}.err_torch_bf16"] = torch.tensor(eb, dtype=torch.float64)


def gen_default64():
    print("default UNet, B=2, 4x64x64 (configs[1] shape)")
    out = {}
    x, ctx = seeded((2, 4, 64, 64), 181), seeded((2, 77, 768), 182)
    t = torch.tensor([431, 37], dtype=torch.long)
    out["t"] = t
    one_case("p", {}, 3, dict(gain=1.0), x, t, ctx, out)
    one_case("w", {}, 3, WELL, x, t, ctx, out)
    save("unet_default64.npz", **out)


def gen_default64_moderate():
    """The same shape on oracle/unet_ref.MODERATE_GATE: a conditioned filler whose GELU gates stay non-linear."""
    print("default UNet, B=2, 4x64x64, moderate-gate filler")
    out = {}
    x, ctx = seeded((2, 4, 64, 64), 181), seeded((2, 77, 768), 182)
    t = torch.tensor([431, 37], dtype=torch.long)
    out["t"] = t
    one_case("m", {}, 3, uref.MODERATE_GATE, x, t, ctx, out)
    small = {}
    xs, cs = seeded((4, 4, 32, 32), 11), seeded((4, 77, 768), 12)
    one_case("small_m", mg.SMALL, 7, uref.MODERATE_GATE, xs, torch.tensor([999, 500, 20, 0], dtype=torch.long), cs, small)
    out.update(small)
    save("unet_default64_moderate.npz", **out)


def gen_default96():
    print("default UNet, B=1, 4x96x96 (configs[4] extent)")
    out = {}
    x, ctx = seeded((1, 4, 96, 96), 191), seeded((1, 77, 768), 192)
    t = torch.tensor([650], dtype=torch.long)
    out["t"] = t
    one_case("w", {}, 3, WELL, x, t, ctx, out)
    save("unet_default96.npz", **out)


def psnr(a, ref):
    ref = ref.double()
    mse = float(((a.double() - ref) ** 2).mean())
    return 10 * math.log10(float(ref.max() - ref.min()) ** 2 / mse)


def gen_wellcond():
    print("well-conditioned filler: small UNet, default@16x16, 50-step trajectory")
    out = {}
    x, ctx = seeded((4, 4, 32, 32), 11), seeded((4, 77, 768), 12)
    t = torch.tensor([999, 500, 20, 0], dtype=torch.long)
    one_case("small", mg.SMALL, 7, WELL, x, t, ctx, out)
    x16, ctx16 = seeded((1, 4, 16, 16), 81), seeded((1, 77, 768), 82)
    one_case("default16", {}, 3, WELL, x16, torch.tensor([431]), ctx16, out)

    # config-1 trajectory on this filler: reference float32, float64 oracle, torch bf16 -- injected per-step noise
    net, cfg, sd = ref_unet(mg.SMALL, 7, **WELL)
    sch = mg.RefScheduler(device=mg.CPU)
    sch.set_timesteps(50, device="cpu")
    tabs = mg.sref.build_tables()
    ts = [int(v) for v in sch.timesteps]
    sd64 = {k: v.double() for k, v in sd.items()}
    sdb = {k: v.bfloat16() for k, v in sd.items()}
    lat = seeded((4, 4, 32, 32), 31) * sch.init_noise_sigma
    lat64, latb = lat.double(), lat.clone()
    real = torch.randn_like
    with torch.no_grad():
        for i, tt in enumerate(sch.timesteps):
            nz = seeded((4, 4, 32, 32), 1000 + i)
            eps = net(sch.scale_model_input(lat, tt), tt, ctx)
            torch.randn_like = lambda m: nz.to(m.dtype)
            try:
                lat = sch.step(eps, tt, lat).prev_sample
            finally:
                torch.randn_like = real
            s = ts[i + 1] if i + 1 < len(ts) else -1
            # float64 chain: float64 UNet, the scheduler's float32 coefficients applied in float64
            eps64 = uref.unet_forward(sd64, cfg, lat64, tt, ctx.double())
            c = [float(v) for v in mg.sref.step_coefficients(tabs, ts[i], s)]
            x0 = (lat64 - c[0] * eps64) / c[1]
            lat64 = c[2] * x0 + c[3] * lat64 + c[4] * nz.double()
            # torch bf16 UNet, float32 scheduler (what autocast does in the reference's sampling loop)
            epsb = uref.unet_forward(sdb, cfg, latb.bfloat16(), tt, ctx.bfloat16()).float()
            latb, _ = mg.sref.step(tabs, epsb, ts[i], s, latb, nz)
            if i in (0, 24, 49):
                out[f"traj.eps.{i}"] = eps
                out[f"traj.lat.{i}"] = lat
            if i + 1 in (24, 49):
                out[f"traj.lat.{i}"] = lat
    out["traj.final"] = lat
    out["traj.final_f64"] = lat64
    p32, pb = psnr(lat, lat64), psnr(latb, lat64)
    print(f"  50-step sample vs float64 chain: reference float32 {p32:.1f} dB, torch bf16 {pb:.1f} dB")
    out["traj.psnr_ref_f32"] = torch.tensor(p32, dtype=torch.float64)
    out["traj.psnr_torch_bf16"] = torch.tensor(pb, dtype=torch.float64)
    save("unet_wellcond.npz", **out)


def gen_scheduler_s5():
    """S5: predict_start_from_noise / predict_noise_from_start / q_posterior / q_posterior_between and the
    module-level helpers extract / get_beta_schedule of the reference, on fixed inputs with per-sample timesteps."""
    print("scheduler S5 methods")
    import ldm.scheduling_ddpm as RS  # the reference module (sys.path set by make_golden)
    sref = mg.sref
    out = {}
    x_t, x0, nz = seeded((5, 4, 8, 8), 201), seeded((5, 4, 8, 8), 202), seeded((5, 4, 8, 8), 203)
    t = torch.tensor([999, 500, 20, 1, 0], dtype=torch.long)
    s = torch.tensor([979, 480, 0, 0, -1], dtype=torch.long)
    out.update(x_t=x_t, x0=x0, noise=nz, t=t, s=s)
    for sched in ("linear", "cosine"):
        ref = mg.RefScheduler(beta_schedule=sched, device=mg.CPU)
        tabs = sref.build_tables(sched, 1000)
        r = {"start": ref.predict_start_from_noise(x_t, t, nz), "noise": ref.predict_noise_from_start(x_t, t, x0)}
        r["qp.mean"], r["qp.std"] = ref.q_posterior(x0, x_t, t)
        r["qb.mean"], r["qb.std"] = ref.q_posterior_between(x0, x_t, t, s)
        m = {"start": sref.predict_start_from_noise(tabs, x_t, t, nz), "noise": sref.predict_noise_from_start(tabs, x_t, t, x0)}
        m["qp.mean"], m["qp.std"] = sref.q_posterior(tabs, x0, x_t, t)
        m["qb.mean"], m["qb.std"] = sref.q_posterior_between(tabs, x0, x_t, t, s)
        for k in r:
            assert torch.equal(r[k], m[k]), (sched, k)
            out[f"{sched}.{k}"] = r[k]
        assert torch.equal(RS.extract(ref.alphas_cumprod, t, x_t.shape), tabs["alphas_cumprod"][t].reshape(5, 1, 1, 1))
    for sched in ("linear", "cosine", "sigmoid"):
        for T in (1000, 250):
            b = RS.get_beta_schedule(sched, T)
            assert b.dtype == torch.float64 and torch.equal(b, sref.betas_f64(sched, T))
            out[f"betas64.{sched}.{T}"] = b
    save("scheduler_s5.npz", **out)


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(8)
    which = sys.argv[1:] or ["s5", "64", "96", "wellcond"]
    if "s5" in which:
        gen_scheduler_s5()
    if "wellcond" in which:
        gen_wellcond()
    if "64" in which:
        gen_default64()
    if "64m" in which:
        gen_default64_moderate()
    if "96" in which:
        gen_default96()
    print("done")
