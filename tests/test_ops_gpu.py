"""GPU parity of every UNet kernel, called through the C-ABI, against a plain float32 torch-CPU evaluation of the
same operator on the same (already bf16-rounded, where applicable) inputs.

Tolerances (relative L2 of the whole output): float32 kernels 2e-5 (exact-fp32 MFMA, different summation order);
bf16 kernels 6e-3 = output rounding to bf16 (2^-9 per element, ~2.3e-3 rel-L2) plus bf16 rounding of P in attention.
The north-star gates (1e-4 fp32 / 2e-2 bf16 per block) sit above these.
"""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2, seeded

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 2e-5, torch.bfloat16: 6e-3}
DTYPES = [torch.bfloat16, torch.float32]


def _ops():
    from ldm import ops
    return ops


def _q(x, dtype):
    """Round to the kernel's storage dtype and return (device tensor, float32 CPU copy of the rounded values)."""
    xd = x.to(dtype)
    return xd.cuda(), xd.float()


def nhwc(x):  # NCHW cpu -> NHWC
    return x.permute(0, 2, 3, 1).contiguous()


# ------------------------------------------------------------------------------------ pack_weight
def test_pack_weight_layouts():
    ops = _ops()
    w = seeded((64, 32, 3, 3), 1)
    p = ops.pack_weight(w.cuda(), torch.float32).cpu()
    assert torch.equal(p, w.permute(0, 2, 3, 1).reshape(64, -1))
    p = ops.pack_weight(w.cuda(), torch.bfloat16).cpu()
    assert torch.equal(p, w.permute(0, 2, 3, 1).reshape(64, -1).bfloat16())
    w2 = seeded((128, 40), 2)
    p = ops.pack_weight(w2.cuda(), torch.float32, glu_pairs=True).cpu()
    ff = 64
    exp = torch.empty_like(w2)
    for blk in range(ff // 16):
        exp[32 * blk:32 * blk + 16] = w2[16 * blk:16 * blk + 16]
        exp[32 * blk + 16:32 * blk + 32] = w2[ff + 16 * blk:ff + 16 * blk + 16]
    assert torch.equal(p, exp)
    b = seeded((128,), 3)
    assert torch.equal(ops.pack_weight(b.cuda(), torch.float32, glu_pairs=True).cpu().flatten(), exp.new_tensor(
        torch.cat([torch.cat([b[16 * k:16 * k + 16], b[ff + 16 * k:ff + 16 * k + 16]]) for k in range(ff // 16)])))


# ------------------------------------------------------------------------------------- conv_gemm
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("M,N,K", [(300, 320, 320), (128, 128, 64), (1000, 640, 768), (77, 1280, 768), (33, 64, 32),
                                   (4100, 160, 96), (256, 2560, 320), (50, 20, 64), (130, 328, 128)])
def test_linear_bias(dtype, M, N, K):
    ops = _ops()
    x, xr = _q(seeded((M, K), 1), dtype)
    w = seeded((N, K), 2) / math.sqrt(K)
    b = seeded((N,), 3)
    wp = ops.pack_weight(w.cuda(), dtype)
    y = ops.conv_gemm(x, wp, bias=b.cuda())
    ref = F.linear(xr, wp.float().cpu(), b)
    assert y.shape == (M, N) and y.dtype == dtype
    assert rel_l2(y, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_epilogues(dtype):
    ops = _ops()
    B, L, K, N = 3, 70, 320, 320
    x, xr = _q(seeded((B, L, K), 1), dtype)
    w = seeded((N, K), 2) / math.sqrt(K)
    b = seeded((N,), 3)
    res, resr = _q(seeded((B, L, N), 4), dtype)
    rv, rvr = _q(seeded((B, N), 5), dtype)
    wp = ops.pack_weight(w.cuda(), dtype)
    wr = wp.float().cpu()
    y = ops.conv_gemm(x, wp, bias=b.cuda(), residual=res)
    assert rel_l2(y, F.linear(xr, wr, b) + resr) < TOL[dtype]
    y = ops.conv_gemm(x, wp, bias=b.cuda(), rowvec=rv)
    assert rel_l2(y, F.linear(xr, wr, b) + rvr[:, None, :]) < TOL[dtype]
    y = ops.conv_gemm(x, wp, bias=b.cuda(), silu=True)
    assert rel_l2(y, F.silu(F.linear(xr, wr, b))) < TOL[dtype]
    y = ops.conv_gemm(x, wp)  # no bias (q_proj / kv_proj)
    assert rel_l2(y, F.linear(xr, wr)) < TOL[dtype]
    y = ops.conv_gemm(x, wp, bias=b.cuda(), out_dtype=torch.float32)
    assert y.dtype == torch.float32 and rel_l2(y, F.linear(xr, wr, b)) < 2e-5


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("act", ["geglu", "swiglu"])
def test_gated_projection(dtype, act):
    """GEGLU / SwiGLU (unet.py:787-790 / :765-768): value = first half, gate = second half of w1_w2."""
    ops = _ops()
    M, C, ff = 200, 64, 256
    x, xr = _q(seeded((M, C), 1), dtype)
    w = seeded((2 * ff, C), 2) / math.sqrt(C)
    b = seeded((2 * ff,), 3)
    wp = ops.pack_weight(w.cuda(), dtype, glu_pairs=True)
    bp = ops.pack_weight(b.cuda(), torch.float32, glu_pairs=True).flatten()
    y = ops.conv_gemm(x, wp, bias=bp, glu=act)
    full = F.linear(xr, w.to(dtype).float(), b)
    val, gate = full.chunk(2, dim=-1)
    ref = val * (F.gelu(gate) if act == "geglu" else F.silu(gate))
    assert y.shape == (M, ff)
    assert rel_l2(y, ref) < TOL[dtype]


def _conv_ref(x_nchw, w, b, stride=1, upsample=False):
    if upsample:
        x_nchw = F.interpolate(x_nchw, scale_factor=2.0, mode="nearest")
    return F.conv2d(x_nchw, w, b, stride=stride, padding=w.shape[-1] // 2)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,Cin,Cout,H,W,stride,up", [
    (2, 64, 128, 16, 16, 1, False), (1, 320, 320, 12, 10, 1, False), (3, 32, 64, 9, 7, 1, False),
    (2, 64, 64, 16, 16, 2, False), (1, 128, 160, 7, 9, 2, False), (2, 64, 64, 8, 8, 1, True), (1, 96, 32, 5, 6, 1, True),
])
def test_conv3x3(dtype, B, Cin, Cout, H, W, stride, up):
    ops = _ops()
    x = seeded((B, Cin, H, W), 1)
    w = seeded((Cout, Cin, 3, 3), 2) / math.sqrt(9 * Cin)
    b = seeded((Cout,), 3)
    xd, xr = _q(nhwc(x), dtype)
    wp = ops.pack_weight(w.cuda(), dtype)
    y = ops.conv_gemm(xd, wp, ksize=3, stride=stride, upsample=up, bias=b.cuda())
    ref = nhwc(_conv_ref(xr.permute(0, 3, 1, 2), w.to(dtype).float(), b, stride, up))
    assert y.shape == ref.shape
    assert rel_l2(y, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("ksize", [1, 3])
def test_conv_two_sources(dtype, ksize):
    """The channel concat of unet.py:313/:377 read in place: [hidden (c0), skip (c1)]."""
    ops = _ops()
    B, c0, c1, H, W, Cout = 2, 128, 64, 8, 8, 128
    a, ar = _q(nhwc(seeded((B, c0, H, W), 1)), dtype)
    s, sr = _q(nhwc(seeded((B, c1, H, W), 2)), dtype)
    w = seeded((Cout, c0 + c1, ksize, ksize), 3) / math.sqrt(ksize * ksize * (c0 + c1))
    b = seeded((Cout,), 4)
    rv, rvr = _q(seeded((B, Cout), 5), dtype)
    wp = ops.pack_weight(w.cuda(), dtype)
    y = ops.conv_gemm(a, wp, a1=s, ksize=ksize, bias=b.cuda(), rowvec=rv)
    cat = torch.cat([ar, sr], dim=-1).permute(0, 3, 1, 2)
    ref = nhwc(F.conv2d(cat, w.to(dtype).float(), b, padding=ksize // 2) + rvr[:, :, None, None])
    assert rel_l2(y, ref) < TOL[dtype]


@pytest.mark.parametrize("tile", ["64,128", "64,160", "128,128", "128,160", "256,128", "256,160"])
def test_every_bf16_tile(tile, monkeypatch):
    """Every BK=64 instantiation of the bf16 kernel (4-wave double-buffered and 8-wave 3-slot-ring), forced through the
    LDM_GEMM_TILE tuning knob so coverage does not depend on the tile heuristic: linear + residual, GLU, 3x3 conv with
    two sources + time embedding + ragged M, stride 2 and upsample."""
    ops = _ops()
    dtype = torch.bfloat16
    monkeypatch.setenv("LDM_GEMM_TILE", tile)
    bm, bn = (int(v) for v in tile.split(","))
    x, xr = _q(seeded((8200, 640), 1), dtype)
    w = seeded((320, 640), 2) / math.sqrt(640)
    b = seeded((320,), 3)
    res, resr = _q(seeded((8200, 320), 4), dtype)
    wp = ops.pack_weight(w.cuda(), dtype)
    y = ops.conv_gemm(x, wp, bias=b.cuda(), residual=res)
    assert rel_l2(y, F.linear(xr, wp.float().cpu(), b) + resr) < TOL[dtype]
    wg = seeded((512, 640), 5) / math.sqrt(640)
    bg = seeded((512,), 6)
    y = ops.conv_gemm(x, ops.pack_weight(wg.cuda(), dtype, glu_pairs=True),
                      bias=ops.pack_weight(bg.cuda(), torch.float32, glu_pairs=True).flatten(), glu="geglu")
    val, gate = F.linear(xr, wg.to(dtype).float(), bg).chunk(2, dim=-1)
    assert rel_l2(y, val * F.gelu(gate)) < TOL[dtype]
    B, c0, c1, H, W, Cout = 9, 64, 64, 31, 30, 160
    a, ar = _q(nhwc(seeded((B, c0, H, W), 7)), dtype)
    s, sr = _q(nhwc(seeded((B, c1, H, W), 8)), dtype)
    wc = seeded((Cout, c0 + c1, 3, 3), 9) / math.sqrt(9 * (c0 + c1))
    bc = seeded((Cout,), 10)
    rv, rvr = _q(seeded((B, Cout), 11), dtype)
    y = ops.conv_gemm(a, ops.pack_weight(wc.cuda(), dtype), a1=s, ksize=3, bias=bc.cuda(), rowvec=rv)
    cat = torch.cat([ar, sr], dim=-1).permute(0, 3, 1, 2)
    ref = nhwc(F.conv2d(cat, wc.to(dtype).float(), bc, padding=1) + rvr[:, :, None, None])
    assert rel_l2(y, ref) < TOL[dtype]
    for stride, up in ((2, False), (1, True)):
        xx, xxr = _q(nhwc(seeded((8, 64, 64, 64), 12)), dtype)
        w3 = seeded((160, 64, 3, 3), 13) / math.sqrt(9 * 64)
        y = ops.conv_gemm(xx, ops.pack_weight(w3.cuda(), dtype), ksize=3, stride=stride, upsample=up, bias=bc.cuda())
        ref = nhwc(_conv_ref(xxr.permute(0, 3, 1, 2), w3.to(dtype).float(), bc, stride, up))
        assert rel_l2(y, ref) < TOL[dtype]
    import ctypes as C
    from ldm import _lib
    d = _lib.ConvGemmDesc(c0=640, c1=0, batch=8200, hin=1, win=1, hout=1, wout=1, ksize=1, stride=1, n=320, dtype=1, out_dtype=1)
    t = (C.c_int * 3)()
    _lib.check(_lib.load().ldm_conv_gemm_tile(C.byref(d), t))
    assert list(t) == [bm, bn, 64], list(t)


@pytest.mark.parametrize("form", ["0", "1"])
def test_ring_forms_of_the_256_row_tiles_agree_bit_for_bit(form, monkeypatch):
    """The 256-row tiles run their 3-slot ring as a ping-pong of two wave groups (default; form 1 = the variant the folded
    upsample uses); LDM_GEMM_PINGPONG=0 is the plain ring.  Same accumulation order per output element -> identical bits:
    3x3 conv with two sources + time embedding + ragged M, stride 2, nearest upsample, 1x1 with residual, short K."""
    ops = _ops()
    dtype = torch.bfloat16
    monkeypatch.setenv("LDM_GEMM_TILE", "256,160")
    B, c0, c1, H, W, Cout = 9, 64, 64, 31, 30, 160
    a, _ = _q(nhwc(seeded((B, c0, H, W), 7)), dtype)
    s2, _ = _q(nhwc(seeded((B, c1, H, W), 8)), dtype)
    wc = ops.pack_weight((seeded((Cout, c0 + c1, 3, 3), 9) / math.sqrt(9 * (c0 + c1))).cuda(), dtype)
    bc = seeded((Cout,), 10).cuda()
    rv, _ = _q(seeded((B, Cout), 11), dtype)
    xx, _ = _q(nhwc(seeded((8, 64, 64, 64), 12)), dtype)
    w3 = ops.pack_weight((seeded((160, 64, 3, 3), 13) / math.sqrt(9 * 64)).cuda(), dtype)
    x, _ = _q(seeded((8200, 640), 1), dtype)
    wl = ops.pack_weight((seeded((320, 640), 2) / math.sqrt(640)).cuda(), dtype)
    res, _ = _q(seeded((8200, 320), 4), dtype)
    xs, _ = _q(seeded((700, 64), 5), dtype)
    wsh = ops.pack_weight(seeded((160, 64), 6).cuda(), dtype)
    x2, _ = _q(seeded((300, 128), 14), dtype)                                # two and three K-tiles: the ring's prologue cases
    w2 = ops.pack_weight(seeded((320, 128), 15).cuda(), dtype)
    x3, _ = _q(seeded((300, 192), 16), dtype)
    w3k = ops.pack_weight(seeded((320, 192), 17).cuda(), dtype)

    def run():
        return [ops.conv_gemm(a, wc, a1=s2, ksize=3, bias=bc, rowvec=rv),
                ops.conv_gemm(xx, w3, ksize=3, stride=2, bias=bc),
                ops.conv_gemm(xx, w3, ksize=3, upsample=True, bias=bc),
                ops.conv_gemm(x, wl, residual=res),
                ops.conv_gemm(xs, wsh),                                   # one K-tile
                ops.conv_gemm(x2, w2), ops.conv_gemm(x3, w3k)]
    ref = run()
    monkeypatch.setenv("LDM_GEMM_PINGPONG", form)
    for got, want in zip(run(), ref):
        assert torch.equal(got, want)


@pytest.mark.parametrize("act", ["geglu", "swiglu"])
@pytest.mark.parametrize("tile", [None, "64,128", "256,128"])
def test_glu_epilogue_keeps_the_preactivation(act, tile, monkeypatch):
    """Training forward of the feed-forward: the GLU epilogue also stores the biased pre-activation (glu_preact).  It must be
    bit-equal to what the plain GEMM writes (same accumulation, same rounding), and the gated output must equal the
    inference epilogue's (the same launch without the second store)."""
    ops = _ops()
    dtype = torch.bfloat16
    if tile:
        monkeypatch.setenv("LDM_GEMM_TILE", tile)
    x, xr = _q(seeded((1000, 320), 1), dtype)
    w = seeded((512, 320), 2) / math.sqrt(320)
    b = seeded((512,), 3)
    wp = ops.pack_weight(w.cuda(), dtype, glu_pairs=True)
    bp = ops.pack_weight(b.cuda(), torch.float32, glu_pairs=True).flatten()
    out, pre = ops.conv_gemm(x, wp, bias=bp, glu=act, glu_preact=True)
    assert out.shape == (1000, 256) and pre.shape == (1000, 512) and pre.dtype == dtype
    assert torch.equal(pre, ops.conv_gemm(x, wp, bias=bp))
    assert torch.equal(out, ops.conv_gemm(x, wp, bias=bp, glu=act))
    val, gate = F.linear(xr, w.to(dtype).float(), b).chunk(2, dim=-1)
    ref = val * (F.gelu(gate) if act == "geglu" else F.silu(gate))
    assert rel_l2(out, ref) < TOL[dtype]
    assert rel_l2(ops.glu_forward(pre, act), ref) < 2 * TOL[dtype]
    with pytest.raises(ValueError):
        ops.conv_gemm(x, wp, bias=bp, glu_preact=True)                      # no GLU epilogue


@pytest.mark.parametrize("tile", [None, "64,128,2", "128,128,3", "128,160,4", "256,128,2", "256,160,3", "64,160,8"])
def test_split_k(tile, monkeypatch):
    """Small-M layers (the 8x8 level) split K over blockIdx.y into a lent workspace and fold deterministically:
    3x3 conv with two sources + time embedding + bias, and linear + bias + SiLU + residual into float32."""
    ops = _ops()
    dtype = torch.bfloat16
    if tile:
        monkeypatch.setenv("LDM_GEMM_TILE", tile)
    B, c0, c1, H, W, Cout = 5, 128, 64, 8, 8, 320
    a, ar = _q(nhwc(seeded((B, c0, H, W), 1)), dtype)
    s_, sr = _q(nhwc(seeded((B, c1, H, W), 2)), dtype)
    w = seeded((Cout, c0 + c1, 3, 3), 3) / math.sqrt(9 * (c0 + c1))
    b = seeded((Cout,), 4)
    rv, rvr = _q(seeded((B, Cout), 5), dtype)
    wp = ops.pack_weight(w.cuda(), dtype)
    y = ops.conv_gemm(a, wp, a1=s_, ksize=3, bias=b.cuda(), rowvec=rv)
    y2 = ops.conv_gemm(a, wp, a1=s_, ksize=3, bias=b.cuda(), rowvec=rv)
    assert torch.equal(y, y2)   # deterministic fold
    cat = torch.cat([ar, sr], dim=-1).permute(0, 3, 1, 2)
    ref = nhwc(F.conv2d(cat, w.to(dtype).float(), b, padding=1) + rvr[:, :, None, None])
    assert rel_l2(y, ref) < TOL[dtype]
    x, xr = _q(seeded((300, 2048), 6), dtype)
    wl = seeded((328, 2048), 7) / math.sqrt(2048)
    res, resr = _q(seeded((300, 328), 8), dtype)
    bl = seeded((328,), 9)
    wlp = ops.pack_weight(wl.cuda(), dtype)
    y = ops.conv_gemm(x, wlp, bias=bl.cuda(), silu=True, residual=res, out_dtype=torch.float32)
    ref = F.silu(F.linear(xr, wlp.float().cpu(), bl)) + resr
    assert y.dtype == torch.float32 and rel_l2(y, ref) < 2e-5 + 1e-3  # SiLU by v_exp/v_rcp: ~1e-6 relative


# ---------------------------------------------------------------------------------------- backward
def _wgrad_ref(x_nchw, dy_nchw, wshape, stride=1, upsample=False):
    """torch autograd of F.conv2d on CPU float32 (inputs already rounded to bf16)."""
    w = torch.zeros(wshape, requires_grad=True)
    xin = F.interpolate(x_nchw, scale_factor=2.0, mode="nearest") if upsample else x_nchw
    y = F.conv2d(xin, w, None, stride=stride, padding=wshape[-1] // 2)
    y.backward(dy_nchw)
    return w.grad


@pytest.mark.parametrize("B,c0,c1,Cout,H,W,ks,stride,up", [
    (2, 64, 0, 128, 16, 16, 3, 1, False), (3, 96, 32, 72, 9, 7, 3, 1, False), (1, 320, 0, 320, 12, 10, 3, 1, False),
    (2, 64, 0, 64, 16, 16, 3, 2, False), (2, 64, 0, 40, 8, 8, 3, 1, True), (4, 128, 64, 136, 8, 8, 1, 1, False),
    (33, 40, 0, 24, 8, 8, 3, 1, False), (2, 64, 0, 320, 8, 8, 3, 1, False), (8, 64, 0, 320, 32, 32, 3, 1, False),
    (2, 320, 0, 64, 16, 16, 1, 1, False), (1, 64, 64, 96, 24, 24, 3, 1, False), (1, 24, 0, 30, 6, 6, 3, 1, False),
    (2, 72, 0, 18, 5, 5, 3, 1, False)])
def test_conv_wgrad(B, c0, c1, Cout, H, W, ks, stride, up):
    """ldm_conv_wgrad against autograd of F.conv2d: 3x3 / 1x1, two sources (two passes), stride 2, upsample, ragged
    tiles, K-tiles that straddle taps, power-of-two and other extents (fast / general staging), every large-tile shape
    (256x256, 320x256, 256x320), enough pixels for several pixel splits, and filter counts / channel counts that leave the
    3x3 fold's last workgroup and last wave ragged."""
    ops = _ops()
    dt = torch.bfloat16
    x, xr = _q(nhwc(seeded((B, c0, H, W), 1)), dt)
    s_, sr = (_q(nhwc(seeded((B, c1, H, W), 2)), dt) if c1 else (None, None))
    Ho = (H * (2 if up else 1) + 2 * (ks // 2) - ks) // stride + 1
    Wo = (W * (2 if up else 1) + 2 * (ks // 2) - ks) // stride + 1
    dy, dyr = _q(nhwc(seeded((B, Cout, Ho, Wo), 3)), dt)
    dw = ops.conv_wgrad(x, dy, a1=s_, ksize=ks, stride=stride, upsample=up)
    xin = xr if s_ is None else torch.cat([xr, sr], dim=-1)
    ref = _wgrad_ref(xin.permute(0, 3, 1, 2), dyr.permute(0, 3, 1, 2), (Cout, c0 + c1, ks, ks), stride, up)
    assert dw.numel() == ref.numel() and dw.dtype == torch.float32   # 1x1 matrices come back as [N, C]
    dw = dw.reshape(ref.shape)
    assert rel_l2(dw, ref) < 2e-5          # bf16 products are exact in fp32; only the summation order differs
    acc = dw.clone().reshape(-1, c0 + c1) if ks == 1 else dw.clone()
    dw2 = ops.conv_wgrad(x, dy, a1=s_, ksize=ks, stride=stride, upsample=up, out=acc)
    assert rel_l2(dw2.reshape(ref.shape), 2 * ref) < 2e-5     # accumulate


def test_linear_wgrad():
    ops = _ops()
    dt = torch.bfloat16
    x, xr = _q(seeded((3, 700, 320), 1), dt)
    dy, dyr = _q(seeded((3, 700, 200), 2), dt)
    dw = ops.conv_wgrad(x, dy)
    ref = dyr.reshape(-1, 200).t() @ xr.reshape(-1, 320)
    assert dw.shape == (200, 320) and rel_l2(dw, ref) < 2e-5


@pytest.mark.parametrize("B,Cin,Cout,H,W,ks,stride,up", [
    (2, 64, 128, 16, 16, 3, 1, False), (1, 96, 32, 9, 7, 3, 1, False), (2, 64, 96, 16, 16, 3, 2, False),
    (1, 32, 64, 7, 9, 3, 2, False), (2, 64, 32, 8, 8, 3, 1, True), (3, 128, 96, 5, 6, 1, 1, False)])
def test_conv_dgrad(B, Cin, Cout, H, W, ks, stride, up):
    """Input gradient through the forward kernel with the transposed / rotated filter: stride 1, stride 2 (zero
    insertion, odd extents) and nearest-upsample (2x2 fold)."""
    ops = _ops()
    dt = torch.bfloat16
    w = seeded((Cout, Cin, ks, ks), 2) / math.sqrt(ks * ks * Cin)
    x = seeded((B, Cin, H, W), 1).to(dt).float().requires_grad_(True)
    wq = w.to(dt).float()
    y = _conv_ref(x, wq, None, stride, up)
    dyr = seeded(tuple(y.shape), 3).to(dt).float()
    y.backward(dyr)
    wd = ops.pack_weight_dgrad(w.cuda(), dt)
    dx = ops.conv_dgrad(nhwc(dyr).to(dt).cuda(), wd, ksize=ks, stride=stride, upsample=up, in_hw=(H, W))
    assert dx.shape == (B, H, W, Cin)
    assert rel_l2(dx, nhwc(x.grad)) < TOL[dt]


def test_colsum_and_sum2x2():
    ops = _ops()
    for dt in DTYPES:
        x, xr = _q(seeded((6, 700, 328), 1), dt)
        assert rel_l2(ops.colsum(x, 6), xr.sum(1)) < 2e-6
        assert rel_l2(ops.colsum(x, 1), xr.sum((0, 1))[None]) < 2e-6
        acc = ops.colsum(x, 6)
        ops.colsum(x, 6, out=acc)
        assert rel_l2(acc, 2 * xr.sum(1)) < 2e-6
        # every column-blocking case of the partial kernel (n / 8 chunks: 1, 40 -> 6 rows per pass, 64, 160 -> 3 z-blocks, 320 -> 5),
        # ragged row counts, more rows than slabs x 64; same bits on a second call (fixed summation order)
        for rows, n in ((3, 8), (4100, 320), (70000, 320), (1000, 512), (900, 1280), (257, 2560)):
            y, yr = _q(seeded((rows, n), 4), dt)
            got = ops.colsum(y, 1)
            assert rel_l2(got, yr.double().sum(0)[None]) < 5e-6, (rows, n)
            assert torch.equal(got, ops.colsum(y, 1))
        u, ur = _q(seeded((2, 6, 10, 40), 2), dt)
        r, rr = _q(seeded((2, 3, 5, 40), 3), dt)
        ref = ur.reshape(2, 3, 2, 5, 2, 40).sum((2, 4)) + rr
        assert rel_l2(ops.sum2x2(u, residual=r), ref) < TOL[dt]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("act", ["geglu", "swiglu"])
def test_glu_training_form(dtype, act):
    """glu_forward on the packed-order pre-activation equals the fused epilogue's result; glu_backward vs autograd."""
    ops = _ops()
    M, ff = 150, 64
    hr = seeded((M, 2 * ff), 1).to(dtype).float()       # checkpoint order: [value | gate]
    blocks = hr.reshape(M, 2, ff // 16, 16).permute(0, 2, 1, 3).reshape(M, 2 * ff)   # packed order
    h = blocks.to(dtype).cuda()
    hv = hr.clone().requires_grad_(True)
    val, gate = hv.chunk(2, dim=-1)
    out_ref = val * (F.gelu(gate) if act == "geglu" else F.silu(gate))
    out = ops.glu_forward(h, act)
    assert rel_l2(out, out_ref.detach()) < TOL[dtype]
    dor = seeded((M, ff), 2).to(dtype).float()
    out_ref.backward(dor)
    dh = ops.glu_backward(h, dor.to(dtype).cuda(), act)
    dref = hv.grad.reshape(M, 2, ff // 16, 16).permute(0, 2, 1, 3).reshape(M, 2 * ff)
    assert rel_l2(dh, dref) < TOL[dtype]
    xs, xsr = _q(seeded((1000,), 3), dtype)
    xg = xsr.clone().requires_grad_(True)
    F.silu(xg).backward(dor.flatten()[:1000])
    assert rel_l2(ops.silu_backward(xs, dor.flatten()[:1000].to(dtype).cuda()), xg.grad) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,c0,c1,H,W,silu", [(2, 320, 0, 16, 16, True), (3, 64, 32, 5, 7, True), (1, 1280, 1280, 8, 8, True),
                                               (2, 64, 0, 12, 12, False)])
def test_groupnorm_backward(dtype, B, c0, c1, H, W, silu):
    ops = _ops()
    C = c0 + c1
    x0, x0r = _q(seeded((B, H, W, c0), 1), dtype)
    x1, x1r = (_q(seeded((B, H, W, c1), 2), dtype) if c1 else (None, None))
    gamma = 1.0 + 0.1 * seeded((C,), 3)
    beta = 0.1 * seeded((C,), 4)
    dy, dyr = _q(seeded((B, H, W, C), 5), dtype)
    add, addr = _q(seeded((B, H, W, C), 6), dtype)
    xr = (x0r if c1 == 0 else torch.cat([x0r, x1r], -1)).clone().requires_grad_(True)
    g = gamma.clone().requires_grad_(True)
    b = beta.clone().requires_grad_(True)
    y = F.group_norm(xr.permute(0, 3, 1, 2), 32, g, b, 1e-5)
    if silu:
        y = F.silu(y)
    y.backward(dyr.permute(0, 3, 1, 2))
    _, stats = ops.groupnorm_with_stats(x0, gamma.cuda(), beta.cuda(), 32, a1=x1, silu=silu)
    dx, dg, db = ops.groupnorm_backward(x0, stats, gamma.cuda(), beta.cuda(), 32, dy, a1=x1, silu=silu, add=add)
    tol = 3 * TOL[dtype] if dtype == torch.bfloat16 else 2e-5
    assert rel_l2(dx, xr.grad + addr) < tol
    assert rel_l2(dg, g.grad) < 2e-5 + (1e-3 if dtype == torch.bfloat16 else 0)
    assert rel_l2(db, b.grad) < 2e-5 + (1e-3 if dtype == torch.bfloat16 else 0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,cols", [(300, 320), (5000, 640), (77, 1280), (3, 64)])
def test_layernorm_backward(dtype, rows, cols):
    ops = _ops()
    x, xr = _q(seeded((rows, cols), 1), dtype)
    gamma = 1.0 + 0.1 * seeded((cols,), 2)
    beta = 0.1 * seeded((cols,), 3)
    dy, dyr = _q(seeded((rows, cols), 4), dtype)
    xg = xr.clone().requires_grad_(True)
    g = gamma.clone().requires_grad_(True)
    b = beta.clone().requires_grad_(True)
    F.layer_norm(xg, (cols,), g, b, 1e-5).backward(dyr)
    dx, dg, db = ops.layernorm_backward(x, gamma.cuda(), dy)
    assert rel_l2(dx, xg.grad) < (2 * TOL[dtype] if dtype == torch.bfloat16 else 2e-5)
    assert rel_l2(dg, g.grad) < 2e-5 and rel_l2(db, b.grad) < 2e-5
    dg2, db2 = dg.clone(), db.clone()
    ops.layernorm_backward(x, gamma.cuda(), dy, dgamma=dg2, dbeta=db2)
    assert rel_l2(dg2, 2 * g.grad) < 2e-5


@pytest.mark.parametrize("B,H,Lq,Lk,d", [(2, 8, 256, 256, 40), (1, 8, 200, 77, 40), (2, 4, 128, 128, 80), (1, 8, 70, 77, 80),
                                         (2, 2, 64, 64, 160), (1, 8, 100, 300, 160), (1, 2, 130, 190, 64),
                                         (1, 2, 70, 45, 40), (2, 2, 257, 129, 40), (1, 1, 33, 17, 40)])
def test_attention_backward(B, H, Lq, Lk, d):
    """Flash backward (dq; dk, dv) against autograd of F.scaled_dot_product_attention in float32 on the same
    bf16-rounded operands; also the lse the training forward saves."""
    ops = _ops()
    dt = torch.bfloat16
    C = H * d
    q, qr = _q(seeded((B, Lq, C), 1), dt)
    kv, kvr = _q(seeded((B, Lk, 2 * C), 2), dt)
    do, dor = _q(seeded((B, Lq, C), 3), dt)
    out, lse = ops.attention_train(q, kv, H)
    assert rel_l2(out, ops.attention(q, kv, H)) == 0.0
    qg = qr.clone().requires_grad_(True)
    kvg = kvr.clone().requires_grad_(True)

    def heads(t):
        return t.reshape(t.shape[0], t.shape[1], H, d).transpose(1, 2)
    k_, v_ = kvg.chunk(2, dim=-1)
    o_ref = F.scaled_dot_product_attention(heads(qg), heads(k_), heads(v_)).transpose(1, 2).reshape(B, Lq, C)
    scores = (heads(qr) @ heads(kvr[..., :C]).transpose(-1, -2)) / math.sqrt(d)
    assert rel_l2(lse, torch.logsumexp(scores, -1) * 1.4426950408889634) < 1e-3
    o_ref.backward(dor)
    dq, dkv = ops.attention_backward(q, kv, out, do, lse, H)
    assert rel_l2(dq, qg.grad) < 2e-2
    assert rel_l2(dkv[..., :C], kvg.grad[..., :C]) < 2e-2
    assert rel_l2(dkv[..., C:], kvg.grad[..., C:]) < 2e-2
    # strided outputs: the column slices of one fused (B, L, 3C) buffer (self-attention)
    if Lq == Lk:
        fused = torch.zeros(B, Lq, 3 * C, dtype=dt, device="cuda")
        ops.attention_backward(q, kv, out, do, lse, H, dq=fused[..., :C], dkv=fused[..., C:])
        assert torch.equal(fused[..., :C], dq) and torch.equal(fused[..., C:], dkv)


def test_large_outputs_stored_with_the_nontemporal_hint_are_the_same_bits(monkeypatch):
    """Outputs of 128 MB and more go out through non-temporal stores (LDM_GEMM_NT_MB): same values, plain and GLU epilogue,
    with the threshold lowered so that a small case takes that path."""
    ops = _ops()
    dt = torch.bfloat16
    x, _ = _q(seeded((2, 16, 16, 320), 1), dt)
    w = seeded((1280, 320), 2) * 0.05
    b = 0.1 * seeded((1280,), 3)
    res, _ = _q(seeded((2, 16, 16, 1280), 4), dt)
    wp = ops.pack_weight(w.cuda(), dt)
    wg = ops.pack_weight(w.cuda(), dt, glu_pairs=True)
    monkeypatch.setenv("LDM_GEMM_NT_MB", "0")
    plain = ops.conv_gemm(x, wp, bias=b.cuda(), residual=res)
    bg = ops.pack_weight(b.cuda(), torch.float32, glu_pairs=True).flatten()
    gated = ops.conv_gemm(x, wg, bias=bg, glu="geglu")
    monkeypatch.setenv("LDM_GEMM_NT_MB", "1")
    plain_nt = ops.conv_gemm(x, wp, bias=b.cuda(), residual=res)
    gated_nt = ops.conv_gemm(x, wg, bias=bg, glu="geglu")
    assert torch.equal(plain, plain_nt) and torch.equal(gated, gated_nt)


def test_conv_gemm_rejects_bad_shapes():
    ops = _ops()
    x = torch.zeros(2, 4, 4, 24, device="cuda", dtype=torch.bfloat16)  # 24 channels: not a multiple of 32
    w = torch.zeros(32, 9 * 24, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(ValueError):
        ops.conv_gemm(x, w, ksize=3)
    with pytest.raises(RuntimeError):
        ops.conv_gemm(x.cpu(), w.cpu(), ksize=3)


# ----------------------------------------------------------------------------------------- norms
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,C,H,W,silu", [(2, 320, 16, 16, True), (3, 64, 5, 7, False), (1, 1280, 8, 8, True),
                                            (2, 32, 32, 32, True)])
def test_groupnorm(dtype, B, C, H, W, silu):
    ops = _ops()
    x, xr = _q(nhwc(seeded((B, C, H, W), 1) * 1.7 + 0.3), dtype)
    g, be = 1 + 0.1 * seeded((C,), 2), 0.1 * seeded((C,), 3)
    y = ops.groupnorm(x, g.cuda(), be.cuda(), 32, silu=silu)
    ref = F.group_norm(xr.permute(0, 3, 1, 2), 32, g, be, 1e-5)
    ref = nhwc(F.silu(ref) if silu else ref)
    assert rel_l2(y, ref) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("c0,c1", [(640, 320), (1280, 640), (64, 32), (1280, 1280)])
def test_groupnorm_concat_straddle(dtype, c0, c1):
    """960 = 640+320 channels in 32 groups of 30: groups straddle the concat boundary."""
    ops = _ops()
    B, H, W = 2, 8, 8
    a, ar = _q(nhwc(seeded((B, c0, H, W), 1)), dtype)
    s, sr = _q(nhwc(seeded((B, c1, H, W), 2) * 2.0 - 0.5), dtype)
    C = c0 + c1
    g, be = 1 + 0.1 * seeded((C,), 3), 0.1 * seeded((C,), 4)
    y = ops.groupnorm(a, g.cuda(), be.cuda(), 32, a1=s, silu=True)
    cat = torch.cat([ar, sr], dim=-1).permute(0, 3, 1, 2)
    ref = nhwc(F.silu(F.group_norm(cat, 32, g, be, 1e-5)))
    assert y.shape == (B, H, W, C)
    assert rel_l2(y, ref) < TOL[dtype]


@pytest.mark.parametrize("B,c0,c1,H,W", [(2, 640, 0, 32, 32), (2, 1280, 640, 32, 32), (2, 640, 320, 32, 32), (3, 1280, 0, 16, 16),
                                          (2, 1280, 1280, 16, 16), (2, 1280, 1280, 8, 8), (1, 320, 0, 24, 24), (2, 96, 32, 7, 9),
                                          (2, 320, 0, 64, 64), (1, 320, 320, 64, 64), (1, 640, 320, 64, 64)])
def test_groupnorm_one_kernel_shapes(B, c0, c1, H, W):
    """ldm_groupnorm on the UNet's bf16 shapes (one workgroup per (sample, group set): 256, 512 and 1024 threads, up to
    16 pixels per thread -- 21 for the 40-channel group sets of the 64x64 level --, groups that straddle a chunk or the
    concat boundary) and the one 64x64 shape (960 = 640 + 320 channels) that stays on the three-kernel path: output against F.group_norm + SiLU on the same bf16-rounded input, and the
    mean / rstd tensor (what the backward reads) against float64 statistics of that input."""
    ops = _ops()
    C = c0 + c1
    a, ar = _q(seeded((B, H, W, c0), 11) * 1.3 + 0.4, torch.bfloat16)
    s, sr = (_q(seeded((B, H, W, c1), 12) * 2.0 - 0.5, torch.bfloat16) if c1 else (None, None))
    g, be = 1 + 0.1 * seeded((C,), 13), 0.1 * seeded((C,), 14)
    y, stats = ops.groupnorm_with_stats(a, g.cuda(), be.cuda(), 32, a1=s, silu=True)
    xr = ar if c1 == 0 else torch.cat([ar, sr], dim=-1)
    ref = nhwc(F.silu(F.group_norm(xr.permute(0, 3, 1, 2), 32, g, be, 1e-5)))
    assert y.shape == (B, H, W, C) and rel_l2(y, ref) < TOL[torch.bfloat16]
    x64 = xr.double().reshape(B, H * W, 32, C // 32)
    mean = x64.mean(dim=(1, 3))
    rstd = 1.0 / torch.sqrt(x64.var(dim=(1, 3), unbiased=False) + 1e-5)
    assert torch.allclose(stats[..., 0].cpu().double(), mean, rtol=0, atol=2e-6 * float(x64.abs().max()))
    assert torch.allclose(stats[..., 1].cpu().double(), rstd, rtol=1e-5, atol=0)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("rows,C", [(1000, 320), (77, 640), (5, 1280), (4096, 32), (3, 64)])
def test_layernorm(dtype, rows, C):
    ops = _ops()
    x, xr = _q(seeded((rows, C), 1) * 2 + 0.5, dtype)
    g, be = 1 + 0.1 * seeded((C,), 2), 0.1 * seeded((C,), 3)
    y = ops.layernorm(x, g.cuda(), be.cuda())
    assert rel_l2(y, F.layer_norm(xr, (C,), g, be, 1e-5)) < TOL[dtype]


# ------------------------------------------------------------------------------------- attention
def _attn_ref(q, kv, heads):
    B, L, C = q.shape
    d = C // heads
    k, v = kv.split(C, dim=-1)

    def sp(z):
        return z.reshape(B, -1, heads, d).permute(0, 2, 1, 3)

    w = torch.softmax(sp(q) @ sp(k).transpose(-1, -2) / math.sqrt(d), dim=-1)
    return (w @ sp(v)).permute(0, 2, 1, 3).reshape(B, L, C)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,L,Lk,C", [
    (2, 256, 256, 320),    # d=40 self, 4 key tiles
    (1, 1024, 1024, 320),  # d=40, many tiles, 8 query blocks
    (2, 200, 77, 320),     # d=40 cross, ragged queries and keys
    (2, 128, 128, 640),    # d=80
    (1, 64, 64, 1280),     # d=160, fewer queries than a workgroup holds
    (2, 64, 77, 1280),     # d=160 cross
    (2, 100, 100, 32),     # d=4  (small config) -> generic path
    (1, 50, 77, 64),       # d=8
])
def test_attention(dtype, B, L, Lk, C):
    ops = _ops()
    q, qr = _q(seeded((B, L, C), 1), dtype)
    kv, kvr = _q(seeded((B, Lk, 2 * C), 2), dtype)
    y = ops.attention(q, kv, 8)
    ref = _attn_ref(qr, kvr, 8)
    assert rel_l2(y, ref) < TOL[dtype]


def test_attention_softmax_extremes():
    """Large score spread: the online-softmax rescale must fire across tiles (one dominant key in a late tile)."""
    ops = _ops()
    B, L, C = 1, 128, 320
    q = seeded((B, L, C), 1)
    kv = seeded((B, 256, 2 * C), 2)
    kv[0, 200, :C] = 6.0 * q[0, 5, :]  # key 200 (4th tile) lines up with query 5 in every head
    qd, qr = _q(q, torch.bfloat16)
    kvd, kvr = _q(kv, torch.bfloat16)
    y = ops.attention(qd, kvd, 8)
    assert rel_l2(y, _attn_ref(qr, kvr, 8)) < TOL[torch.bfloat16]
    assert torch.isfinite(y.float()).all()


# ---------------------------------------------------------------------------- first / last layers
@pytest.mark.parametrize("dtype", DTYPES)
def test_timestep_embedding(dtype):
    ops = _ops()
    from oracle.unet_ref import timestep_embedding
    t = torch.tensor([0.0, 1.0, 20.0, 500.0, 999.0, 37.5])
    y = ops.timestep_embedding(t.cuda(), 320, dtype)
    ref = timestep_embedding(t, 320)
    # sin/cos of arguments up to ~1e3: one float32 ulp of the argument is 6e-5
    assert (y.float().cpu() - ref).abs().max() < (3e-4 if dtype == torch.float32 else 5e-3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("B,H,W,C", [(2, 16, 16, 320), (1, 7, 9, 32), (2, 64, 64, 320), (2, 48, 52, 64), (1, 66, 66, 320),
                                     (4, 128, 128, 64), (1, 262, 254, 32)])   # >= 65,536 pixels: the 16-lanes-per-pixel final conv
def test_conv_in_out(dtype, B, H, W, C):
    ops = _ops()
    x = seeded((B, 4, H, W), 1)
    w = seeded((C, 4, 3, 3), 2) / 6.0
    b = seeded((C,), 3)
    xd, xr = _q(x, dtype)
    y = ops.conv_in(xd, w.cuda(), b.cuda(), dtype)
    assert rel_l2(y, nhwc(F.conv2d(xr, w, b, padding=1))) < TOL[dtype]
    y32 = ops.conv_in(x.cuda(), w.cuda(), b.cuda(), dtype)  # float32 latents into a bf16 net
    assert rel_l2(y32, nhwc(F.conv2d(x, w, b, padding=1))) < TOL[dtype]

    h = seeded((B, C, H, W), 4)
    w2 = seeded((4, C, 3, 3), 5) / math.sqrt(9 * C)
    b2 = seeded((4,), 6)
    hd, hr = _q(nhwc(h), dtype)
    z = ops.conv_out(hd, w2.cuda(), b2.cuda(), dtype)
    assert z.shape == (B, 4, H, W)
    assert rel_l2(z, F.conv2d(hr.permute(0, 3, 1, 2), w2, b2, padding=1)) < TOL[dtype]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("shape,glu", [((64, 96, 3, 3), False), ((40, 24, 3, 3), False), ((320, 768), False), ((128, 40), True),
                                       ((33, 7, 3, 3), False)])
def test_pack_weight_pair_matches_the_two_single_packs(dtype, shape, glu):
    """ldm_pack_weight_pair (one pass, both layouts) against ldm_pack_weight of the filter and of its transposed / rotated
    form, bit for bit; ragged extents; column ranges of a fused dgrad buffer."""
    ops = _ops()
    from ldm._train import glu_rows_to_packed
    w = seeded(shape, 5).cuda()
    fwd, dg = ops.pack_weight_pair(w, dtype, glu_pairs=glu)
    assert torch.equal(fwd, ops.pack_weight(w, dtype, glu_pairs=glu))
    assert torch.equal(dg, ops.pack_weight_dgrad(glu_rows_to_packed(w) if glu else w, dtype))
    if len(shape) == 2 and not glu:
        w2 = seeded((shape[0] + 32, shape[1]), 6).cuda()
        rows = shape[0] + w2.shape[0]
        f = torch.zeros(rows, shape[1], dtype=dtype, device="cuda")
        d = torch.zeros(shape[1], rows, dtype=dtype, device="cuda")
        ops.pack_weight_pair(w, dtype, fwd=f[:shape[0]], dgrad=d[:, :shape[0]])
        ops.pack_weight_pair(w2, dtype, fwd=f[shape[0]:], dgrad=d[:, shape[0]:])
        cat = torch.cat([w, w2], 0)
        assert torch.equal(f, ops.pack_weight(cat, dtype)) and torch.equal(d, ops.pack_weight_dgrad(cat, dtype))


# ------------------------------------------------------------------- the BENCHMARKED shapes (round 2)
def _attn_ref_by_head(qr, kvr, heads, dout=None):
    """float32 torch-CPU softmax(QK^T/sqrt d)V, one head at a time (the 9,216-token score matrix of one head is 340 MB);
    with dout also the gradients (dq, dkv) by autograd, head by head."""
    B, L, C = qr.shape
    d = C // heads
    out = torch.empty_like(qr)
    dq = torch.zeros_like(qr) if dout is not None else None
    dkv = torch.zeros_like(kvr) if dout is not None else None
    for b in range(B):
        for h in range(heads):
            sl = slice(h * d, (h + 1) * d)
            q = qr[b, :, sl].clone().requires_grad_(dout is not None)
            k = kvr[b, :, sl].clone().requires_grad_(dout is not None)
            v = kvr[b, :, C + h * d:C + (h + 1) * d].clone().requires_grad_(dout is not None)
            o = torch.softmax(q @ k.t() / math.sqrt(d), dim=-1) @ v
            out[b, :, sl] = o.detach()
            if dout is not None:
                o.backward(dout[b, :, sl])
                dq[b, :, sl] = q.grad
                dkv[b, :, sl] = k.grad
                dkv[b, :, C + h * d:C + (h + 1) * d] = v.grad
    return out, dq, dkv


@pytest.mark.parametrize("L,Lk", [(4096, 4096), (9216, 9216), (4096, 77)])
def test_attention_headline_lengths_d40(L, Lk):
    """d = 40 self-attention at the token counts that are benchmarked: 4,096 (64x64 latents, 16 % of the configs[1] step,
    unet.py:678) and 9,216 (96x96, configs[4]); forward (inference and training form + lse) and backward, bf16, against
    float32 torch on the same bf16-rounded operands."""
    ops = _ops()
    B, C, heads = 1, 320, 8
    q, qr = _q(seeded((B, L, C), 1), torch.bfloat16)
    kv, kvr = _q(seeded((B, Lk, 2 * C), 2), torch.bfloat16)
    do, dor = _q(seeded((B, L, C), 3), torch.bfloat16)
    ref, dq_ref, dkv_ref = _attn_ref_by_head(qr, kvr, heads, dor)
    y = ops.attention(q, kv, heads)
    assert rel_l2(y, ref) < TOL[torch.bfloat16]
    y2, lse = ops.attention_train(q, kv, heads)
    assert torch.equal(y2, y)
    d = C // heads
    s0 = (qr[0, :, :d] @ kvr[0, :, :d].t()) * (math.log2(math.e) / math.sqrt(d))
    # d = 40: the kernel folds log2(e)/sqrt(d) into a bf16-rounded copy of Q, so its scores carry 2^-9 relative rounding
    assert torch.allclose(lse[0, 0].cpu(), torch.logsumexp(s0 * math.log(2.0), dim=-1) / math.log(2.0), atol=3e-2, rtol=0)
    dq, dkv = ops.attention_backward(q, kv, y2, do, lse, heads)
    assert rel_l2(dq, dq_ref) < 1e-2 and rel_l2(dkv, dkv_ref) < 1e-2


def test_conv3x3_headline_rows():
    """The implicit-GEMM 3x3 conv at configs[1]'s largest row count: 320 -> 320 channels, 64x64, B = 32, i.e. M = 131,072
    rows x K = 2,880 (unet.py:702/:708 at level 0), bf16, with bias + time-embedding row vector + residual; checked against
    F.conv2d on three whole samples of the batch (first, middle, last: every tile row band and the M edge)."""
    ops = _ops()
    B, C, H = 32, 320, 64
    x, xr = _q(nhwc(seeded((B, C, H, H), 1)), torch.bfloat16)
    w = seeded((C, C, 3, 3), 2) / math.sqrt(9 * C)
    b = seeded((C,), 3)
    rv, rvr = _q(seeded((B, C), 4), torch.bfloat16)
    res, resr = _q(nhwc(seeded((B, C, H, H), 5)), torch.bfloat16)
    wp = ops.pack_weight(w.cuda(), torch.bfloat16)
    y = ops.conv_gemm(x, wp, ksize=3, bias=b.cuda(), rowvec=rv, residual=res)
    assert y.shape == (B, H, H, C)
    wr = w.bfloat16().float()
    for s in (0, 13, 31):
        ref = F.conv2d(xr[s:s + 1].permute(0, 3, 1, 2), wr, b, padding=1) + rvr[s][None, :, None, None]
        ref = nhwc(ref) + resr[s:s + 1]
        assert rel_l2(y[s:s + 1], ref) < TOL[torch.bfloat16], s
    # the K = 320 projections of the same level (q|k|v: N = 960; GEGLU: N = 2560) on the same 131,072 rows
    xm = x.view(B * H * H, C)
    wq = seeded((960, C), 6) / math.sqrt(C)
    yq = ops.conv_gemm(xm, ops.pack_weight(wq.cuda(), torch.bfloat16))
    rows = torch.tensor([0, 1, 4095, 4096, 70000, 131071])
    assert rel_l2(yq[rows.cuda()], F.linear(xr.view(-1, C)[rows], wq.bfloat16().float())) < TOL[torch.bfloat16]
    wg = seeded((2560, C), 7) / math.sqrt(C)
    bg = seeded((2560,), 8)
    yg = ops.conv_gemm(xm, ops.pack_weight(wg.cuda(), torch.bfloat16, glu_pairs=True),
                       bias=ops.pack_weight(bg.cuda(), torch.float32, glu_pairs=True).flatten(), glu="geglu")
    full = F.linear(xr.view(-1, C)[rows], wg.bfloat16().float(), bg)
    val, gate = full.chunk(2, dim=-1)
    assert rel_l2(yg[rows.cuda()], val * F.gelu(gate)) < TOL[torch.bfloat16]


@pytest.mark.parametrize("M,C,N,glu", [(300, 320, 960, None), (1000, 640, 640, None), (130, 1280, 1280, None),
                                       (257, 320, 2560, "geglu"), (64, 640, 5120, "swiglu"), (4096, 320, 320, None)])
def test_layernorm_folded_into_gemms(M, C, N, glu):
    """LDM_EPI_ROWSTATS + LDM_EPI_LNFOLD (unet.py:550-551, :576 without a LayerNorm pass): a producer GEMM with bias + residual
    writes x and its per-row partial statistics; the consumer GEMM runs on x with the gamma-scaled filter and finishes
    rstd (acc - mean colsum) + bias' in its epilogue.  Against float32 torch: LayerNorm of the stored bf16 x, then the linear
    layer (and GEGLU / SwiGLU) -- same tolerance as the unfused kernels."""
    ops = _ops()
    a, ar = _q(seeded((M, C), 1), torch.bfloat16)
    res, resr = _q(seeded((M, C), 2) * 0.5 + 0.3, torch.bfloat16)        # a mean that is not zero
    wp_ = seeded((C, C), 3) / math.sqrt(C)
    bp = seeded((C,), 4)
    x, stats = ops.conv_gemm(a, ops.pack_weight(wp_.cuda(), torch.bfloat16), bias=bp.cuda(), residual=res, rowstats=True)
    assert stats.shape == (M, C // 80, 2)
    xr = x.float().cpu()
    assert rel_l2(x, F.linear(ar, wp_.bfloat16().float(), bp) + resr) < TOL[torch.bfloat16]
    tot = stats.sum(dim=1).cpu().double()
    assert torch.allclose(tot[:, 0], xr.double().sum(1), rtol=0, atol=1e-3)
    assert torch.allclose(tot[:, 1], (xr.double() ** 2).sum(1), rtol=1e-5, atol=1e-3)
    gamma, beta = 1 + 0.1 * seeded((C,), 5), 0.1 * seeded((C,), 6)
    nrows = 2 * N if glu else N
    w = seeded((nrows, C), 7) / math.sqrt(C)
    b = seeded((nrows,), 8)
    wf, colsum, bf = ops.fold_layernorm(w.cuda(), b.cuda(), gamma.cuda(), beta.cuda(), torch.bfloat16, glu_pairs=bool(glu))
    y = ops.conv_gemm(x, wf, bias=bf, glu=glu, ln=(stats, colsum, 1e-5))
    full = F.linear(F.layer_norm(xr, (C,), gamma, beta, 1e-5), w, b)
    if glu:
        val, gate = full.chunk(2, dim=-1)
        full = val * (F.gelu(gate) if glu == "geglu" else F.silu(gate))
    assert y.shape == (M, N)
    assert rel_l2(y, full) < TOL[torch.bfloat16]
    # and against the unfused kernels on the same x: LayerNorm kernel + plain GEMM
    y0 = ops.conv_gemm(ops.layernorm(x, gamma.cuda(), beta.cuda()), ops.pack_weight(w.cuda(), torch.bfloat16, glu_pairs=bool(glu)),
                       bias=ops.pack_weight(b.cuda(), torch.float32, glu_pairs=True).flatten() if glu else b.cuda(), glu=glu)
    assert rel_l2(y, y0) < 2 * TOL[torch.bfloat16]
    with pytest.raises(ValueError):
        ops.conv_gemm(a, ops.pack_weight(seeded((96, C), 9).cuda(), torch.bfloat16), rowstats=True)   # N not a multiple of 160
