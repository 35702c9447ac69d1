"""The kernel SOURCES of latent-diffusion_amd/csrc, compiled for the host against the SIMT emulator (tests/emu), run on CPU
memory with PROT_NONE pages behind every buffer and behind the LDS image -- the checks a GPU cannot give without a fault:

  * an access one byte past a buffer, or past the dynamic LDS a launch asked for, is a SIGSEGV with the lane that did it;
  * raw buffer loads follow the range-check rule LLVM documents (the scalar offset is NOT checked), so zero-fill that only
    works when the hardware also checks the scalar offset shows up as garbage;
  * LDS-DMA completes either as late as the counted `s_waitcnt vmcnt(n)` allow or at issue: a missing wait reads stale LDS
    in the first model, a ring slot refilled too early corrupts operands in the second.

These are a bounded sample (about two minutes); the whole -m gpu suite runs the same way with `LDM_EMU=1 python -m pytest
tests -m gpu -n 7` (tests/conftest.py).  The emulator is test infrastructure: the product loads only libldm_amd.so.
Reference values: float32 torch on bf16-rounded operands, as in the GPU tests."""
import math
import os
import subprocess
import sys

import pytest
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emu"))

from conftest import rel_l2, seeded  # noqa: E402
from harness import Guarded, emulated  # noqa: E402

bf = torch.bfloat16


@pytest.fixture(autouse=True)
def _no_forced_tile():
    saved = os.environ.pop("LDM_GEMM_TILE", None)
    yield
    os.environ.pop("LDM_GEMM_TILE", None)
    if saved is not None:
        os.environ["LDM_GEMM_TILE"] = saved


def test_emulator_exports_the_c_abi():
    from harness import emu_lib
    from ldm import _lib
    lib = emu_lib()
    for name in _lib.EXPORTED_SYMBOLS:
        assert hasattr(lib, name), name


@pytest.mark.parametrize("dma", ["lazy", "eager"])
@pytest.mark.parametrize("tile", ["64,128", "64,160", "128,128", "128,160", "256,128", "256,160"])
def test_every_bf16_tile_ragged(tile, dma):
    """Every BK = 64 tile instantiation of the implicit GEMM (the 3-slot ring of the 256-row tiles included), M and N not
    multiples of the tile, bias + residual epilogue, guarded operands and output, both DMA completion models."""
    os.environ["LDM_GEMM_TILE"] = tile
    M, K, N = 300, 192, 200
    with Guarded() as g, emulated(dma, guard=True) as ops:
        a, res = g.like(seeded((M, K), 1).to(bf)), g.like(seeded((M, N), 2).to(bf))
        w, b = seeded((N, K), 3) / math.sqrt(K), g.like(seeded((N,), 4))
        wp = g.like(ops.pack_weight(w, bf))
        y = ops.conv_gemm(a, wp, bias=b, residual=res).clone()
        ref = F.linear(a.float(), w.to(bf).float(), b) + res.float()
    assert rel_l2(y, ref) < 6e-3


@pytest.mark.parametrize("dma", ["lazy", "eager"])
def test_conv3x3_two_sources_stride_upsample(dma):
    """3x3 taps over two concatenated sources (BK = 32 path: the K-tile straddles neither), stride 2 and the folded nearest
    2x upsample, padding taps zero-filled by the descriptor's range check."""
    with Guarded() as g, emulated(dma, guard=True) as ops:
        x0, x1 = g.like(seeded((2, 6, 5, 32), 5).to(bf)), g.like(seeded((2, 6, 5, 64), 6).to(bf))
        w = seeded((48, 96, 3, 3), 7) / math.sqrt(96 * 9)
        b = g.like(seeded((48,), 8))
        wp = g.like(ops.pack_weight(w, bf))
        xin = torch.cat([x0, x1], -1).float().permute(0, 3, 1, 2)
        wr = w.to(bf).float()
        y = ops.conv_gemm(x0, wp, a1=x1, ksize=3, bias=b).clone()
        assert rel_l2(y.permute(0, 3, 1, 2), F.conv2d(xin, wr, b, padding=1)) < 6e-3
        y = ops.conv_gemm(x0, wp, a1=x1, ksize=3, stride=2, bias=b).clone()
        assert rel_l2(y.permute(0, 3, 1, 2), F.conv2d(xin, wr, b, padding=1, stride=2)) < 6e-3
        y = ops.conv_gemm(x0, wp, a1=x1, ksize=3, upsample=True, bias=b).clone()
        assert rel_l2(y.permute(0, 3, 1, 2), F.conv2d(F.interpolate(xin, scale_factor=2.0, mode="nearest"), wr, b, padding=1)) < 6e-3


@pytest.mark.parametrize("ptile", ["64,160", "128,160", "256,160"])
def test_layernorm_fold_on_every_tile(ptile):
    """LayerNorm folded into the GEMMs around it (LDM_EPI_ROWSTATS producer, LDM_EPI_LNFOLD consumers, plain / GEGLU /
    SwiGLU) on every tile that can run them, ragged M, everything on guard pages.  Reference: float32 LayerNorm + linear
    on the producer's bf16 output."""
    M, C = 300, 320
    with Guarded() as g, emulated("lazy", guard=True) as ops:
        a, res = g.like(seeded((M, C), 11).to(bf)), g.like((seeded((M, C), 12) * 0.5 + 0.3).to(bf))
        wp_, bp = seeded((C, C), 13) / math.sqrt(C), g.like(seeded((C,), 14))
        gamma, beta = 1 + 0.1 * seeded((C,), 15), 0.1 * seeded((C,), 16)
        os.environ["LDM_GEMM_TILE"] = ptile
        x, st = ops.conv_gemm(a, g.like(ops.pack_weight(wp_, bf)), bias=bp, residual=res, rowstats=True)
        assert st.shape == (M, C // 80, 2)
        assert rel_l2(x, F.linear(a.float(), wp_.to(bf).float(), bp) + res.float()) < 6e-3
        tot = st.sum(dim=1)
        assert torch.allclose(tot[:, 0], x.float().sum(1), rtol=0, atol=2e-3)
        assert torch.allclose(tot[:, 1], (x.float() ** 2).sum(1), rtol=2e-5, atol=2e-3)
        for N, glu, tiles in ((3 * C, None, ("64,128", "128,160", "256,128", "256,160")), (4 * C, "geglu", ("128,128",)),
                              (4 * C, "swiglu", ("128,128",))):
            w, b = seeded((N, C), 17) / math.sqrt(C), seeded((N,), 18)
            wf, cs, b2 = ops.fold_layernorm(w, b, gamma, beta, bf, glu_pairs=bool(glu))
            wf, cs, b2 = g.like(wf), g.like(cs), g.like(b2)
            full = F.linear(F.layer_norm(x.float(), (C,), gamma, beta, 1e-5), w, b)
            if glu:
                val, gate = full.chunk(2, dim=-1)
                full = val * (F.gelu(gate) if glu == "geglu" else F.silu(gate))
            for tile in tiles:
                os.environ["LDM_GEMM_TILE"] = tile
                y = ops.conv_gemm(x, wf, bias=b2, glu=glu, ln=(st, cs, 1e-5))
                assert rel_l2(y, full) < 6e-3, (N, glu, tile)


def _ref_attention(q, kv, heads):
    B, L, C = q.shape
    d = C // heads
    k, v = kv[..., :C], kv[..., C:]
    qh = q.float().view(B, L, heads, d).transpose(1, 2)
    kh = k.float().reshape(B, -1, heads, d).transpose(1, 2)
    vh = v.float().reshape(B, -1, heads, d).transpose(1, 2)
    return F.scaled_dot_product_attention(qh, kh, vh).transpose(1, 2).reshape(B, L, C)


@pytest.mark.parametrize("v1", [False, True])
@pytest.mark.parametrize("L,Lk,C", [(100, 77, 320), (64, 200, 320), (70, 77, 640)])
def test_attention_ragged_key_tiles_read_nothing_past_the_end(L, Lk, C, v1):
    """Keys in a last tile of fewer than 64 rows, K/V of the LAST sample ending on a guard page: the rows past lk must come
    back as zeros from the buffer range check (strict reading: scalar offsets unchecked), not from memory behind the
    tensor.  Both d = 40 kernels (32x32x16 pipelined and, with LDM_ATTN_V1, 16x16x32) and d = 80."""
    env = dict(os.environ, PYTHONPATH=os.pathsep.join(sys.path))
    if v1:
        env["LDM_ATTN_V1"] = "1"
    code = f"""
import sys, torch
sys.path.insert(0, {os.path.join(HERE, 'emu')!r}); sys.path.insert(0, {HERE!r})
from conftest import seeded, rel_l2
from harness import Guarded, emulated
from test_emu_kernels_cpu import _ref_attention
bf = torch.bfloat16
with Guarded() as g, emulated("lazy", guard=True) as ops:
    q, kv = g.like(seeded((2, {L}, {C}), 21).to(bf)), g.like(seeded((2, {Lk}, {2 * C}), 22).to(bf))
    y = ops.attention(q, kv, 8).clone()
    e = rel_l2(y, _ref_attention(q, kv, 8))
assert e < 6e-3, e
print("ok", e)
"""
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-1500:])


def test_guard_pages_catch_the_round2_fault():
    """The fault that closed GPU access in round 2: a bf16 bias buffer read as float32 by the GEMM epilogue (twice the bytes).
    With the bias on a guard page the emulator dies with SIGSEGV inside the kernel; the product's Python front end now
    refuses such a bias before any launch (second half)."""
    code = f"""
import sys, ctypes as C, torch
sys.path.insert(0, {os.path.join(HERE, 'emu')!r}); sys.path.insert(0, {HERE!r})
import conftest
from harness import Guarded, emulated, emu_lib
from ldm import _lib
bf = torch.bfloat16
g = Guarded()
M, K, N = 64, 64, 2048
a, w = g.like(torch.randn(M, K).to(bf)), g.like(torch.randn(N, K).to(bf))
bias = g.like(torch.randn(N).to(bf))            # N bf16 = half the bytes the kernel will read
out = g.empty((M, N), bf)
d = _lib.ConvGemmDesc(a0=a.data_ptr(), c0=K, c1=0, batch=M, hin=1, win=1, hout=1, wout=1, ksize=1, stride=1, upsample=0,
                      w=w.data_ptr(), n=N, epilogue=_lib.EPI_BIAS, bias=bias.data_ptr(), out=out.data_ptr(), dtype=_lib.BF16,
                      out_dtype=_lib.BF16)
print("launching", flush=True)
emu_lib().ldm_conv_gemm(C.byref(d), None)
print("survived", flush=True)
"""
    env = dict(os.environ, PYTHONPATH=os.pathsep.join(sys.path))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert "launching" in r.stdout and "survived" not in r.stdout and r.returncode == -11, (r.returncode, r.stdout, r.stderr[-800:])
    with emulated() as ops:
        a, w = torch.randn(64, 64).to(bf), torch.randn(2048, 64).to(bf)
        with pytest.raises(TypeError):
            ops.conv_gemm(a, w, bias=torch.randn(2048).to(bf))


def test_transformer_model_with_and_without_the_fold(golden):
    """TransformerModel (GroupNorm -> proj_in -> block -> proj_out) through the product's own schedule: fp32 kernels against the
    golden fixture, bf16 with the LayerNorm fold against bf16 without it and against the fixture."""
    from ldm import ops as _ops
    from ldm import unet as U
    from oracle import unet_ref as uref
    g = golden("blocks.npz")

    def fill(mod, prefix, seed):
        shapes = {f"{prefix}.{k}": tuple(v.shape) for k, v in mod.state_dict().items()}
        sd = uref.fill_state_dict(shapes, seed=seed, gain=1.0)
        mod.load_state_dict({k[len(prefix) + 1:]: v for k, v in sd.items()})
        return mod.eval()

    x, ctx = seeded((1, 640, 8, 8), 71), seeded((1, 77, 768), 62)
    saved = (_ops._ln_fold_checked.copy(), U._LN_FOLD)
    try:
        with emulated("lazy", guard=True), torch.no_grad():
            net = fill(U.TransformerModel(640, 8, 1, 768), "tmodel", 70)
            assert rel_l2(net(x, ctx), g["tmodel.out"]) < 1e-4
            _ops._ln_fold_checked[-1] = True           # the self-check's large case is covered by the tile tests above
            U._LN_FOLD = True
            y_fold = net(x.to(bf), ctx.to(bf)).clone()
            assert net._packed_for(bf).get("ln_fold") is True
            # the entry that faulted the GPU in round 2: float32 module and inputs inside a bf16 autocast region -- the filters
            # are packed INSIDE the region, where torch's own ops return bf16
            net.invalidate_packed_weights()
            with torch.autocast("cuda", dtype=bf):
                assert (torch.ones(2, 2) @ torch.ones(2, 2)).dtype == bf        # the region is live for torch's ops too
                y_auto = net(x, ctx).clone()
            P = net._packed_for(bf)
            assert P.get("ln_fold") is True and all(v.dtype == torch.float32 for k, v in P.items()
                                                    if k.endswith(".b_ln") or k.endswith(".s1"))
            assert y_auto.dtype == bf and torch.equal(y_auto, y_fold)
            net.invalidate_packed_weights()
            U._LN_FOLD = False
            y_plain = net(x.to(bf), ctx.to(bf)).clone()
            assert not net._packed_for(bf).get("ln_fold")
    finally:
        _ops._ln_fold_checked.clear()
        _ops._ln_fold_checked.update(saved[0])
        U._LN_FOLD = saved[1]
    assert rel_l2(y_fold, g["tmodel.out"]) < 2e-2 and rel_l2(y_plain, g["tmodel.out"]) < 2e-2
    assert rel_l2(y_fold, y_plain) < 1.5e-2


def test_smoke_on_the_emulator():
    """__graft_entry__.smoke() with the emulator in place of cuda:0: one denoise step of the small UNet, float32 and bf16, and
    the fused scheduler step, against the oracle -- the whole product path (Python schedule, packing, every forward kernel
    the small net uses) end to end on the host, buffers on guard pages."""
    from ldm.scheduling_ddpm import DDPMScheduler
    from ldm.unet import UNet
    from oracle import scheduler_ref as sref
    from oracle import unet_ref as uref
    kw = dict(block_out_channels=[32, 64, 64, 64])
    cfg = uref.UNetConfig(**kw)
    sd = uref.fill_state_dict(uref.param_shapes(cfg), seed=7, **uref.WELL_CONDITIONED)
    net = UNet(**kw)
    net.load_state_dict(sd)
    net.eval()
    x, ctx, noise = seeded((2, 4, 8, 8), 11), seeded((2, 77, 768), 12), seeded((2, 4, 8, 8), 13)
    t = 500
    with torch.no_grad():
        ref64 = uref.unet_forward({k: v.double() for k, v in sd.items()}, cfg, x.double(), t, ctx.double())
        tb = uref.unet_forward({k: v.bfloat16() for k, v in sd.items()}, cfg, x.bfloat16(), t, ctx.bfloat16()).double()
        with emulated("lazy", guard=True):
            sch = DDPMScheduler(device=torch.device("cpu"))
            sch.set_timesteps(50)
            eps = net(x, t, ctx).clone()
            epsb = net(x.bfloat16(), t, ctx.bfloat16()).clone()
            prev = sch.step(eps, 500, x, noise=noise).prev_sample.clone()
    assert rel_l2(eps, ref64) < 1e-4
    bound = max(2e-2, 1.25 * rel_l2(tb, ref64))
    assert rel_l2(epsb, ref64) < bound, (rel_l2(epsb, ref64), bound)
    ref_prev, _ = sref.step(sref.build_tables(), eps, 500, 480, x, noise)
    assert torch.equal(prev, ref_prev), "scheduler step is not bit-exact on the emulator"


def test_backward_and_small_kernels_through_the_gpu_tests_bodies():
    """The bodies of 71 `-m gpu` operator tests, unchanged, on the emulator (LDM_EMU=1, tests/conftest.py): weight gradient
    (3x3, strided, two sources, linear), input gradient, column sums, the GLU training form, GroupNorm / LayerNorm backward,
    attention backward with a ragged 77-key tile, GroupNorm over a concat whose group straddles the sources, the timestep
    embedding, conv_in / final conv, split-K.  In a child process: the session-wide patches of that mode stay out of this one."""
    sel = ("conv_wgrad or linear_wgrad or conv_dgrad or colsum or glu_training or groupnorm_backward or layernorm_backward or "
           "(attention_backward and 77) or concat_straddle or timestep_embedding or conv_in_out or split_k")
    env = dict(os.environ, LDM_EMU="1")
    env.pop("LDM_GEMM_TILE", None)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_ops_gpu.py"), "-m", "gpu", "-q", "-x",
                        "-p", "no:cacheprovider", "-k", sel], env=env, capture_output=True, text=True, timeout=1500,
                       cwd=os.path.dirname(HERE))
    tail = r.stdout[-1500:]
    assert r.returncode == 0, tail + r.stderr[-1500:]
    assert " passed" in tail and "failed" not in tail, tail
