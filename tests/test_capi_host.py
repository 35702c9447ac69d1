"""CPU: the C-ABI library loads, exports every symbol include/ldm_amd.h declares, and its host-side
scheduler functions (tables, timesteps, step coefficients) are bit-exact with the golden vectors."""
import ctypes as C
import os
import re

import pytest
import torch

from conftest import ROOT, same_bits
from ldm import _lib
from ldm.scheduling_ddpm import DDPMScheduler
from oracle import scheduler_ref as sref


def test_library_exports_header_symbols():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "ldm_amd.h")).read()
    declared = set(re.findall(r"\b(ldm_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/ldm_amd.h but not exported"
    assert declared == set(_lib.EXPORTED_SYMBOLS)
    assert lib.ldm_abi_version() == _lib.ABI_VERSION == 3


@pytest.mark.parametrize("sched", ["linear", "cosine", "sigmoid"])
def test_tables_bit_exact(golden, sched):
    g = golden("scheduler.npz")
    s = DDPMScheduler(beta_schedule=sched, device=torch.device("cpu"))
    for k in ("betas", "alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
              "posterior_mean_coef1", "posterior_mean_coef2", "posterior_std"):
        assert same_bits(getattr(s, k), g[f"{sched}.{k}"]), (sched, k)
    assert s.timesteps.tolist() == list(range(999, -1, -1))
    assert s.config.num_train_timesteps == 1000 and s.config.prediction_type == "epsilon"
    assert s.init_noise_sigma == 1.0


def _ulp_diff(a, b):
    return (a.view(torch.int32).long() - b.view(torch.int32).long()).abs()


@pytest.mark.parametrize("sched,code", [("linear", 0), ("cosine", 1), ("sigmoid", 2)])
def test_c_table_builder(golden, sched, code):
    """ldm_ddpm_build_tables (for non-Python callers): betas and alphas_cumprod bit-exact; the five tables that
    go through sqrt are IEEE-exact and therefore within a few ulp of the torch+MKL results (see scheduling_ddpm.py)."""
    g = golden("scheduler.npz")
    lib = _lib.load()
    names = ("betas", "alphas_cumprod", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
             "posterior_mean_coef1", "posterior_mean_coef2", "posterior_std")
    bufs = [torch.empty(1000, dtype=torch.float32) for _ in names]
    _lib.check(lib.ldm_ddpm_build_tables(code, 1000, *[C.cast(b.data_ptr(), _lib.PF) for b in bufs]))
    for n, b in zip(names, bufs):
        ref = g[f"{sched}.{n}"]
        if n in ("betas", "alphas_cumprod"):
            assert same_bits(b, ref), (sched, n)
        else:
            d = _ulp_diff(b, ref)
            assert int(d.max()) <= 4 and float((d > 0).float().mean()) < 0.05, (sched, n, int(d.max()))
    assert lib.ldm_ddpm_build_tables(7, 1000, *[C.cast(b.data_ptr(), _lib.PF) for b in bufs]) == -1


def test_c_step_coefficients():
    lib = _lib.load()
    tabs = sref.build_tables()
    ptr = [C.cast(tabs[k].data_ptr(), _lib.PF) for k in ("alphas_cumprod", "sqrt_alphas_cumprod",
                                                         "sqrt_one_minus_alphas_cumprod")]
    out = (C.c_float * 5)()
    for t, s in ((980, 960), (500, 480), (20, 0), (0, -1), (999, 998)):
        _lib.check(lib.ldm_ddpm_step_coefficients(*ptr, 1000, t, s, out))
        ref = sref.step_coefficients(tabs, t, s)
        for a, b in zip(list(out), ref):
            assert abs(a - b) <= 2.4e-7 * abs(b), (t, s, a, b)
    assert lib.ldm_ddpm_step_coefficients(*ptr, 1000, 1000, 0, out) == -1


def test_tables_other_lengths():
    for T in (1, 10, 250, 4000):
        for sched in ("linear", "cosine", "sigmoid"):
            s = DDPMScheduler(num_train_timesteps=T, beta_schedule=sched, device=torch.device("cpu"))
            ref = sref.build_tables(sched, T)
            assert same_bits(s.alphas_cumprod, ref["alphas_cumprod"]), (T, sched)
            assert same_bits(s.posterior_std, ref["posterior_std"]), (T, sched)


@pytest.mark.parametrize("spacing", ["leading", "linspace", "trailing"])
def test_set_timesteps(golden, spacing):
    g = golden("scheduler.npz")
    s = DDPMScheduler(timestep_spacing=spacing, device=torch.device("cpu"))
    for n in (50, 10, 3, 1000, 7):
        s.set_timesteps(n, device="cpu")
        assert s.timesteps.dtype == torch.long
        assert same_bits(s.timesteps, g[f"timesteps.{spacing}.{n}"]), (spacing, n)
        assert s.num_inference_steps == n
    for n in list(range(1, 130)) + [250, 333, 500, 999]:  # sweep against the oracle
        ref = sref.inference_timesteps(n, 1000, spacing)
        if ref.numel() != n:
            # 'trailing' quirk: float step rounding gives the reference n+1 entries ending in timestep -1
            # (its sampling loop then fails on the index); the C routine refuses such n.
            assert spacing == "trailing" and int(ref[-1]) == -1
            with pytest.raises(ValueError):
                s.set_timesteps(n)
            continue
        s.set_timesteps(n)
        assert same_bits(s.timesteps, ref), (spacing, n)
    with pytest.raises(ValueError):
        s.set_timesteps(1001)


def test_timesteps_offset_and_bad_spacing(golden):
    g = golden("scheduler.npz")
    s = DDPMScheduler(steps_offset=1, device=torch.device("cpu"))
    s.set_timesteps(20)
    assert same_bits(s.timesteps, g["timesteps.leading.20.offset1"])
    s = DDPMScheduler(timestep_spacing="bogus", device=torch.device("cpu"))
    with pytest.raises(ValueError):
        s.set_timesteps(10)
    with pytest.raises(ValueError):
        DDPMScheduler(beta_schedule="bogus", device=torch.device("cpu"))


def test_previous_timestep_and_coefficients():
    s = DDPMScheduler(device=torch.device("cpu"))
    assert s.previous_timestep(777) == 776
    s.set_timesteps(50)
    assert s.previous_timestep(980) == 960 and s.previous_timestep(0) == -1
    assert int(s.previous_timestep(torch.tensor(20))) == 0
    tabs = sref.build_tables()
    for t, p in ((980, 960), (500, 480), (20, 0), (0, -1), (999, 998), (1, 0)):
        mine = list(s.step_coefficients(t, p))
        ref = sref.step_coefficients(tabs, t, p)
        assert mine == list(ref), (t, p, mine, ref)
    with pytest.raises(ValueError):
        s.step_coefficients(1000, 0)


def test_hot_path_refuses_cpu_tensors():
    s = DDPMScheduler(device=torch.device("cpu"))
    x = torch.zeros(1, 4, 8, 8)
    with pytest.raises(RuntimeError):
        s.step(x, 10, x)
    with pytest.raises(RuntimeError):
        s.add_noise(x, x, torch.zeros(1, dtype=torch.long))
