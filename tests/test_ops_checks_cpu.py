"""The wrong-operand fault class (round 2: a bf16 bias read as float32 ran past its buffer and faulted the GPU), closed at every
front end of ldm/ops.py and not only at conv_gemm: each kernel entry reads its side operands (norm affine parameters, biases,
raw filters, statistics) as dense float32 arrays of an implied length and its activations in ONE dtype, so every front end
must refuse anything else BEFORE a pointer reaches the library.  Runs on the host emulator harness (tests/emu), which only
patches the "is this a GPU tensor" check; `emu_launch_count` proves that a refused call launched nothing.

Second half: pack_tree for the UNet / a 320-channel TransformerModel (LayerNorm fold terms) / the VAE decoder INSIDE a bf16
autocast region -- the entry that faulted -- must produce float32 side operands and compute-dtype filters, entry by entry."""
import os
import sys

import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emu"))

from harness import emu_lib, emulated  # noqa: E402

bf = torch.bfloat16


def _refused(fn, exc=(TypeError, ValueError)):
    lib = emu_lib()
    before = lib.emu_launch_count()
    with pytest.raises(exc):
        fn()
    assert lib.emu_launch_count() == before, "a refused call must not launch a kernel"


def test_norm_front_ends_refuse_wrong_side_operands():
    with emulated() as ops:
        x = torch.randn(2, 16, 64).to(bf)
        g32, b32 = torch.ones(64), torch.zeros(64)
        ops.groupnorm(x, g32, b32, 32)                                       # the good call works
        ops.layernorm(x, g32, b32)
        for bad in (g32.to(bf), torch.ones(32), torch.ones(128)[::2]):
            _refused(lambda: ops.groupnorm(x, bad, b32, 32))
            _refused(lambda: ops.groupnorm(x, g32, bad, 32))
            _refused(lambda: ops.groupnorm_with_stats(x, bad, b32, 32))
            _refused(lambda: ops.layernorm(x, bad, b32))
            _refused(lambda: ops.layernorm(x, g32, bad))
        _refused(lambda: ops.groupnorm(x, g32, b32, 32, a1=torch.randn(2, 16, 64)))           # second source in another dtype
        _refused(lambda: ops.groupnorm(x, torch.ones(128), torch.zeros(128), 32, a1=torch.randn(2, 8, 64).to(bf)))  # other extent
        _refused(lambda: ops.groupnorm(x.transpose(0, 1), g32, b32, 32))                      # not contiguous
        _refused(lambda: ops.groupnorm(x, g32, b32, 5))                                       # 64 channels, 5 groups
        _refused(lambda: ops.layernorm(x[:, :, ::2], torch.ones(32), torch.zeros(32)))
        # backward front ends
        xs, st = ops.groupnorm_with_stats(x, g32, b32, 32)
        dy = torch.randn(2, 16, 64).to(bf)
        ops.groupnorm_backward(x, st, g32, b32, 32, dy)
        _refused(lambda: ops.groupnorm_backward(x, st.to(bf), g32, b32, 32, dy))
        _refused(lambda: ops.groupnorm_backward(x, st[:, :16], g32, b32, 32, dy))
        _refused(lambda: ops.groupnorm_backward(x, st, g32.to(bf), b32, 32, dy))
        _refused(lambda: ops.groupnorm_backward(x, st, g32, b32, 32, dy.float()))
        _refused(lambda: ops.groupnorm_backward(x, st, g32, b32, 32, dy, dgamma=torch.zeros(64).to(bf), dbeta=torch.zeros(64)))
        ops.layernorm_backward(x, g32, dy)
        _refused(lambda: ops.layernorm_backward(x, g32.to(bf), dy))
        _refused(lambda: ops.layernorm_backward(x, g32, dy.float()))
        _refused(lambda: ops.layernorm_backward(x, g32, dy, dgamma=torch.zeros(32), dbeta=torch.zeros(32)))


def test_attention_front_ends_refuse_mismatched_operands():
    with emulated() as ops:
        B, L, Lk, C = 1, 32, 24, 64
        q, kv = torch.randn(B, L, C).to(bf), torch.randn(B, Lk, 2 * C).to(bf)
        ops.attention(q, kv, 8)
        qkv = torch.randn(B, L, 3 * C).to(bf)
        ops.attention(qkv[..., :C], qkv[..., C:], 8)                       # column slices of the fused buffer are fine
        _refused(lambda: ops.attention(q, kv.float(), 8))                  # K|V in another dtype than q
        _refused(lambda: ops.attention(q.float(), kv, 8))
        _refused(lambda: ops.attention(q, torch.randn(B, 2 * C, Lk).to(bf).transpose(1, 2), 8))      # column stride != 1
        _refused(lambda: ops.attention(q, torch.randn(2, Lk, 2 * C).to(bf), 8))                      # other batch
        _refused(lambda: ops.attention_train(q, kv.float(), 8))
        out, lse = ops.attention_train(q, kv, 8)
        do = torch.randn_like(out)
        ops.attention_backward(q, kv, out, do, lse, 8)
        _refused(lambda: ops.attention_backward(q, kv, out, do.float(), lse, 8))
        _refused(lambda: ops.attention_backward(q, kv.float(), out, do, lse, 8))
        _refused(lambda: ops.attention_backward(q, kv, out, do, lse.to(bf), 8))
        _refused(lambda: ops.attention_backward(q, kv, out, do, lse[:, :4], 8))


def test_conv_front_ends_refuse_wrong_side_operands():
    with emulated() as ops:
        x = torch.randn(1, 4, 8, 8)
        w, b = torch.randn(32, 4, 3, 3), torch.randn(32)
        h = ops.conv_in(x, w, b, bf)
        _refused(lambda: ops.conv_in(x, w.to(bf), b, bf))
        _refused(lambda: ops.conv_in(x, w, b.to(bf), bf))
        _refused(lambda: ops.conv_in(x, w, torch.randn(16), bf))
        _refused(lambda: ops.conv_in(x, torch.randn(32, 8, 3, 3), b, bf))
        wo, bo = torch.randn(4, 32, 3, 3), torch.randn(4)
        ops.conv_out(h, wo, bo, bf)
        _refused(lambda: ops.conv_out(h, wo.to(bf), bo, bf))
        _refused(lambda: ops.conv_out(h, wo, bo.double(), bf))
        _refused(lambda: ops.conv_out(h.permute(0, 2, 1, 3), wo, bo, bf))
        # conv_gemm: the checks round 2 added, plus residual / rowvec / LayerNorm-fold operands
        a = torch.randn(64, 64).to(bf)
        wp = ops.pack_weight(torch.randn(160, 64), bf)
        ops.conv_gemm(a, wp, bias=torch.zeros(160))
        _refused(lambda: ops.conv_gemm(a, wp, bias=torch.zeros(160).to(bf)))
        _refused(lambda: ops.conv_gemm(a, wp, bias=torch.zeros(80)))
        _refused(lambda: ops.conv_gemm(a, wp, residual=torch.zeros(64, 160)))                 # float32 residual
        _refused(lambda: ops.conv_gemm(a, wp, residual=torch.zeros(32, 160).to(bf)))          # too few rows
        _refused(lambda: ops.conv_gemm(a.view(4, 16, 64), wp, rowvec=torch.zeros(2, 160).to(bf)))
        _refused(lambda: ops.conv_gemm(a, wp.float()))
        _refused(lambda: ops.conv_gemm(a, wp.t()))
        a320 = torch.randn(64, 320).to(bf)
        w320 = ops.pack_weight(torch.randn(320, 320), bf)
        y, st = ops.conv_gemm(a320, w320, rowstats=True)
        ops.conv_gemm(y, w320, bias=torch.zeros(320), ln=(st, torch.zeros(320), 1e-5))
        _refused(lambda: ops.conv_gemm(y, w320, bias=torch.zeros(320), ln=(st[:, :2].contiguous(), torch.zeros(320), 1e-5)))
        _refused(lambda: ops.conv_gemm(y, w320, bias=torch.zeros(320), ln=(st, torch.zeros(320).to(bf), 1e-5)))
        _refused(lambda: ops.conv_gemm(y, w320, bias=torch.zeros(320), ln=(st, torch.zeros(160), 1e-5)))
        # the C entry refuses a segment count that does not cover c0 as well (a caller that bypasses ops.py)
        import ctypes as C
        from ldm import _lib
        lib = emu_lib()
        out = torch.empty(64, 320, dtype=bf)
        d = _lib.ConvGemmDesc(a0=y.data_ptr(), c0=320, c1=0, batch=64, hin=1, win=1, hout=1, wout=1, ksize=1, stride=1, upsample=0,
                              w=w320.data_ptr(), n=320, epilogue=_lib.EPI_LNFOLD, out=out.data_ptr(), dtype=_lib.BF16,
                              out_dtype=_lib.BF16, ln_stats=st.data_ptr(), ln_colsum=torch.zeros(320).data_ptr(), ln_segments=3,
                              ln_eps=1e-5)
        assert lib.ldm_conv_gemm(C.byref(d), 0) == -1 and b"segments" in lib.ldm_last_error()


def _check_packed(P, dtype, raw_prefixes):
    n_side = n_filt = 0
    for k, v in P.items():
        if not torch.is_tensor(v):
            continue
        assert v.is_contiguous() or k.endswith(".wd"), k
        if k.endswith((".b", ".g", ".s1", ".b_ln")) or any(k.startswith(r) for r in raw_prefixes):
            assert v.dtype == torch.float32, (k, v.dtype)
            n_side += 1
        elif k.endswith((".w", ".wd", ".w_ln")):
            assert v.dtype == dtype, (k, v.dtype)
            n_filt += 1
        else:
            raise AssertionError(f"unclassified packed entry {k}")
    return n_side, n_filt


def test_pack_tree_under_bf16_autocast_keeps_every_side_operand_float32(monkeypatch):
    from ldm import ops as real_ops
    from ldm import unet as U
    from ldm import vae as V
    with emulated():
        monkeypatch.setattr(real_ops, "ln_fold_selfcheck", lambda device: True)     # the fold's runtime self-check has its own test
        net = U.UNet(block_out_channels=[32, 64, 64, 64])
        tm = U.TransformerModel(in_channels=320, num_attention_heads=8)
        vae = V.AutoencoderKL(block_out_channels=(32, 32, 64, 64), norm_num_groups=8)
        with torch.autocast("cuda", dtype=bf):
            assert torch.is_autocast_enabled("cuda")
            for train in (False, True):
                side, filt = _check_packed(U.pack_tree(net, bf, train=train), bf, ("conv_in.", "final_conv.2."))
                assert side > 100 and filt > 100
            P = U.pack_tree(tm, bf)
            assert P.get("ln_fold") is True and "transformer_blocks.0.attn1.qkv.s1" in P
            _check_packed(P, bf, ())
            side, filt = _check_packed(U.pack_tree(vae, bf), bf, ("decoder.conv_in.", "decoder.conv_out.", "post_quant_conv."))
            assert side > 20 and filt > 20
