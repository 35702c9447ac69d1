"""CPU: the Python kernel schedule of a TransformerModel (ldm/unet.py: pack_tree + run_transformer_model), with and without the
LayerNorm fold, executed against torch EMULATIONS of the C-ABI kernels (bf16 storage, float32 accumulation, the GLU row
interleave, the row-statistics layout).  No kernel runs here -- the GPU parity tests do that; this pins the host glue the
kernels are driven by: key names of the packed set, which GEMM produces statistics for which consumer, that bias' / colsum
travel in the packed order, and that both schedules reproduce the oracle (unet.py:507-521, :593-596)."""
import math

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2, seeded
from ldm import ops
from ldm import unet as U
from oracle import unet_ref as uref
from test_ln_fold_cpu import _emu_conv_gemm, _emu_pack_weight

BF = torch.bfloat16


def _pack_weight(w, dtype, glu_pairs=False):
    w = w.detach()
    if w.dim() == 4:                                   # OIHW -> [O][kh][kw][I]
        w = w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)
    return _emu_pack_weight(w, dtype, glu_pairs)


def _pack_weight_pair(w, dtype, *, glu_pairs=False, fwd=None, dgrad=None, want_dgrad=True):
    p = _pack_weight(w, dtype, glu_pairs)
    if fwd is None:
        return p, None
    fwd.copy_(p)
    return fwd, None


def _conv_gemm(a0, w, **kw):
    lead = a0.shape[:-1]
    res = kw.pop("residual", None)
    out = _emu_conv_gemm(a0.reshape(-1, a0.shape[-1]), w, residual=None if res is None else res.reshape(-1, res.shape[-1]), **kw)
    if isinstance(out, tuple):
        return out[0].view(*lead, -1), out[1]
    return out.view(*lead, -1)


def _groupnorm(a0, gamma, beta, groups, *, a1=None, silu=False, eps=1e-5):
    x = a0 if a1 is None else torch.cat([a0, a1], dim=-1)
    y = F.group_norm(x.float().permute(0, 3, 1, 2), groups, gamma, beta, eps).permute(0, 2, 3, 1)
    return (F.silu(y) if silu else y).to(a0.dtype).contiguous()


def _attention(q, kv, heads):
    B, L, C = q.shape
    d = C // heads
    k, v = kv[..., :C].float(), kv[..., C:].float()
    sp = lambda z: z.reshape(B, -1, heads, d).permute(0, 2, 1, 3)
    w = torch.softmax(sp(q.float()) @ sp(k).transpose(-1, -2) / math.sqrt(d), dim=-1)
    return (w @ sp(v)).permute(0, 2, 1, 3).reshape(B, L, C).to(q.dtype)


@pytest.fixture
def emulated_ops(monkeypatch):
    monkeypatch.setattr(ops, "pack_weight", _pack_weight)
    monkeypatch.setattr(ops, "pack_weight_pair", _pack_weight_pair)
    monkeypatch.setattr(ops, "conv_gemm", _conv_gemm)
    monkeypatch.setattr(ops, "groupnorm", _groupnorm)
    monkeypatch.setattr(ops, "layernorm", lambda x, g, b, eps=1e-5: F.layer_norm(x.float(), x.shape[-1:], g, b, eps).to(x.dtype))
    monkeypatch.setattr(ops, "attention", _attention)
    monkeypatch.setattr(ops, "ln_fold_selfcheck", lambda device: True)
    calls = {"layernorm": 0, "rowstats": 0, "ln": 0}
    real_ln, real_gemm = ops.layernorm, ops.conv_gemm

    def counting_ln(*a, **k):
        calls["layernorm"] += 1
        return real_ln(*a, **k)

    def counting_gemm(*a, **k):
        calls["rowstats"] += bool(k.get("rowstats"))
        calls["ln"] += k.get("ln") is not None
        return real_gemm(*a, **k)
    monkeypatch.setattr(ops, "layernorm", counting_ln)
    monkeypatch.setattr(ops, "conv_gemm", counting_gemm)
    return calls


@pytest.mark.parametrize("C,layers,act", [(320, 1, "geglu"), (640, 2, "swiglu")])
def test_transformer_model_schedule_with_and_without_fold(emulated_ops, monkeypatch, C, layers, act):
    cfg = uref.UNetConfig(transformer_layers_per_block=layers, activation_fn=act)
    mod = U.TransformerModel(C, 8, layers, 768, activation_fn=act)
    shapes = {f"tm.{k}": tuple(v.shape) for k, v in mod.state_dict().items()}
    sd = uref.fill_state_dict(shapes, seed=70, gain=1.0)
    mod.load_state_dict({k[3:]: v for k, v in sd.items()})
    x = seeded((2, C, 4, 4), 71)
    ctx = seeded((2, 77, 768), 72)
    with torch.no_grad():
        ref = uref.transformer_model(sd, "tm", x, ctx, cfg)
        h = x.permute(0, 2, 3, 1).contiguous().to(BF)
        outs = {}
        for fold in (True, False):
            monkeypatch.setattr(U, "_LN_FOLD", fold)
            for k in emulated_ops:
                emulated_ops[k] = 0
            P = U.pack_tree(mod, BF)
            assert bool(P.get("ln_fold")) == fold
            y = U.run_transformer_model(P, "", mod, h, ctx.to(BF))
            outs[fold] = y.float().permute(0, 3, 1, 2)
            if fold:      # no LayerNorm pass; every LayerNorm input comes with statistics; three folded GEMMs per block
                assert emulated_ops == {"layernorm": 0, "rowstats": 3 * layers, "ln": 3 * layers}
                for b in range(layers):
                    for key in (f"transformer_blocks.{b}.attn1.qkv", f"transformer_blocks.{b}.attn2.q_proj",
                                f"transformer_blocks.{b}.feed_forward.1.w1_w2"):
                        assert P[key + ".w_ln"].dtype == BF and P[key + ".s1"].dtype == P[key + ".b_ln"].dtype == torch.float32
            else:
                assert emulated_ops == {"layernorm": 3 * layers, "rowstats": 0, "ln": 0}
    assert rel_l2(outs[True], ref) < 2e-2 and rel_l2(outs[False], ref) < 2e-2
    assert rel_l2(outs[True], outs[False]) < 1.5e-2


def test_small_channel_counts_keep_the_layernorm_kernel(emulated_ops):
    mod = U.TransformerModel(64, 8, 1, 768)
    P = U.pack_tree(mod, BF)
    assert "ln_fold" not in P and not any(k.endswith(".w_ln") for k in P)
    P32 = U.pack_tree(U.TransformerModel(320, 8, 1, 768), torch.float32)
    assert "ln_fold" not in P32          # float32 is the exact parity path: never folded
