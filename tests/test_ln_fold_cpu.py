"""CPU: the algebra of the LayerNorm fold (include/ldm_amd.h LDM_EPI_ROWSTATS / LDM_EPI_LNFOLD), without the kernels.

ldm.ops.fold_layernorm_terms produces (W' = W gamma, colsum, bias' = bias + W beta); the GEMM epilogue then evaluates
rstd (x W'^T - mean colsum) + bias' from per-row (sum, sum of squares) partials over 80-column segments.  Here the same
steps run in torch on the host -- bf16-rounded x and W', float32 accumulation -- against LayerNorm + Linear in float64, and
the packing is exercised under an active bf16 autocast region (a bf16 bias handed to the float32 bias pointer is how this
path once faulted on the GPU)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2, seeded
from ldm import ops


def _emulate(x_bf16, w, bias, gamma, beta, eps=1e-5):
    ws, colsum, b2 = ops.fold_layernorm_terms(w, bias, gamma, beta, torch.bfloat16)
    C = x_bf16.shape[1]
    xf = x_bf16.float()
    seg = xf.view(xf.shape[0], C // 80, 80)
    stats = torch.stack([seg.sum(-1), (seg * seg).sum(-1)], dim=-1)           # what LDM_EPI_ROWSTATS writes: (M, C/80, 2)
    sx, sq = stats[..., 0].sum(1), stats[..., 1].sum(1)
    mu = sx / C
    rstd = torch.rsqrt((sq / C - mu * mu).clamp(min=0) + eps)
    acc = xf @ ws.bfloat16().float().t()                                      # the matrix core: bf16 operands, float32 sums
    return rstd[:, None] * (acc - mu[:, None] * colsum[None, :]) + b2[None, :], (ws, colsum, b2)


@pytest.mark.parametrize("M,C,N,shift", [(64, 320, 960, 0.0), (33, 640, 640, 0.7), (16, 1280, 320, -1.5)])
def test_fold_identity(M, C, N, shift):
    x = (seeded((M, C), 1) * 1.3 + shift).bfloat16()
    w = seeded((N, C), 2) / C ** 0.5
    b = seeded((N,), 3)
    gamma, beta = 1 + 0.1 * seeded((C,), 4), 0.1 * seeded((C,), 5)
    y, (ws, colsum, b2) = _emulate(x, w, b, gamma, beta)
    ref = F.linear(F.layer_norm(x.double(), (C,), gamma.double(), beta.double(), 1e-5), w.double(), b.double())
    assert ws.dtype == colsum.dtype == b2.dtype == torch.float32
    # bf16 filter rounding (2^-9 relative per weight) is the only approximation left: well inside the kernels' 6e-3
    assert rel_l2(y, ref) < 4e-3
    y0, _ = _emulate(x, w, None, gamma, beta)
    assert torch.allclose(y - y0, b[None, :].expand_as(y), atol=1e-5)


def test_fold_terms_stay_float32_under_autocast():
    C, N = 320, 160
    w, b = seeded((N, C), 6) / C ** 0.5, seeded((N,), 7)
    gamma, beta = 1 + 0.1 * seeded((C,), 8), 0.1 * seeded((C,), 9)
    ref = ops.fold_layernorm_terms(w, b, gamma, beta, torch.bfloat16)
    with torch.autocast("cpu", dtype=torch.bfloat16):
        got = ops.fold_layernorm_terms(w, b, gamma, beta, torch.bfloat16)
        assert (w @ beta).dtype == torch.bfloat16          # what a plain matmul would have produced in this region
    for a, r in zip(got, ref):
        assert a.dtype == torch.float32 and torch.equal(a, r)


def test_fold_support_predicate_and_validation():
    assert ops.ln_fold_supported(320, torch.bfloat16) and ops.ln_fold_supported(1280, torch.bfloat16)
    assert not ops.ln_fold_supported(320, torch.float32)
    for c in (32, 64, 160, 480):                     # not multiples of lcm(160, 64): unfused LayerNorm kernel
        assert not ops.ln_fold_supported(c, torch.bfloat16)


# ------------------------------------------------------------------ the runtime self-check, executed against host emulations
def _emu_pack_weight(w, dtype, glu_pairs=False):
    w = w.detach().to(torch.float32)
    flat = w.dim() == 1
    if flat:
        w = w[:, None]
    if glu_pairs:      # value rows 16 b .. 16 b + 15, then their gate rows (include/ldm_amd.h ldm_pack_weight)
        n = w.shape[0] // 2
        w = torch.stack([w[:n].view(n // 16, 16, -1), w[n:].view(n // 16, 16, -1)], dim=1).reshape(2 * n, -1)
    return w.to(dtype)


def _emu_conv_gemm(a0, w, *, bias=None, residual=None, glu=None, rowstats=False, ln=None, **_):
    acc = a0.float() @ w.float().t()
    if ln is not None:
        stats, colsum, eps = ln
        K = a0.shape[-1]
        mu = stats[..., 0].sum(1) / K
        rstd = torch.rsqrt((stats[..., 1].sum(1) / K - mu * mu).clamp(min=0) + eps)
        acc = rstd[:, None] * (acc - mu[:, None] * colsum[None, :])
    if bias is not None:
        assert bias.dtype == torch.float32 and bias.numel() == w.shape[0]
        acc = acc + bias[None, :]
    if glu:
        blocks = acc.view(acc.shape[0], -1, 2, 16)
        val, gate = blocks[:, :, 0], blocks[:, :, 1]
        acc = (val * (F.gelu(gate) if glu == "geglu" else F.silu(gate))).reshape(acc.shape[0], -1)
    if residual is not None:
        acc = acc + residual.float()
    out = acc.to(a0.dtype)
    if rowstats:
        seg = out.float().view(out.shape[0], -1, 80)
        return out, torch.stack([seg.sum(-1), (seg * seg).sum(-1)], dim=-1).contiguous()
    return out


def test_selfcheck_runs_and_detects_a_broken_fold(monkeypatch):
    """ops.ln_fold_selfcheck line by line on the host: pack_weight / conv_gemm / layernorm replaced by torch emulations of
    what the kernels compute (including the GLU row interleave of filter, colsum and bias).  A correct fold passes; a fold
    whose column sums are wrong is detected, disabled and reported with a warning."""
    import contextlib
    import warnings
    monkeypatch.setattr(torch.cuda, "is_current_stream_capturing", lambda: False)
    monkeypatch.setattr(torch.cuda, "device", lambda d: contextlib.nullcontext())
    monkeypatch.setattr(ops, "pack_weight", _emu_pack_weight)
    monkeypatch.setattr(ops, "conv_gemm", _emu_conv_gemm)
    monkeypatch.setattr(ops, "layernorm", lambda x, g, b, eps=1e-5: F.layer_norm(x.float(), x.shape[-1:], g, b, eps).to(x.dtype))
    monkeypatch.setattr(ops, "_ln_fold_checked", {})
    cpu = torch.device("cpu")
    with warnings.catch_warnings():
        warnings.simplefilter("error")           # a passing check must not warn
        assert ops.ln_fold_selfcheck(cpu) is True
    assert ops.ln_fold_selfcheck(cpu) is True      # cached

    def broken(a0, w, **kw):
        if kw.get("ln") is not None:
            st, cs, eps = kw["ln"]
            kw["ln"] = (st, cs * 1.5, eps)
        return _emu_conv_gemm(a0, w, **kw)
    monkeypatch.setattr(ops, "conv_gemm", broken)
    monkeypatch.setattr(ops, "_ln_fold_checked", {})
    with pytest.warns(UserWarning, match="self-check did not pass"):
        assert ops.ln_fold_selfcheck(cpu) is False
