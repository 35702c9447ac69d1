"""GPU parity of the VAE-decoder leg against oracle/vae_ref.py.

PARITY UNPINNED: the decoder's arithmetic lives in third-party diffusers (absent offline, unpinned in the reference) and the
reference holds no fixture of its outputs, so the checker is this build's own restatement of the published architecture,
evaluated in float64 with deterministic random weights.  Gates are the UNet's: 1e-4 relative L2 in fp32, 2e-2 in bf16;
the uint8 conversion is bit-exact against torch's own evaluation of the reference's expression.
"""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_l2, seeded
from oracle import vae_ref as vref

pytestmark = pytest.mark.gpu

SMALL = dict(block_out_channels=(32, 64, 64, 64))
TOL = {torch.float32: 1e-4, torch.bfloat16: 2e-2}
DTYPES = [torch.float32, torch.bfloat16]


def _vae(cfg_kw, seed=11):
    from ldm.vae import AutoencoderKL
    cfg = vref.VAEDecoderConfig(**cfg_kw)
    sd = vref.fill_state_dict(vref.param_shapes(cfg), seed=seed, gain=1.0)
    m = AutoencoderKL(**cfg_kw)
    m.load_state_dict(sd)
    return m.cuda().eval(), cfg, sd


def _f64(sd):
    return {k: v.double() for k, v in sd.items()}


# ------------------------------------------------------------------------------------------- kernels
@pytest.mark.parametrize("out_dtype", DTYPES)
@pytest.mark.parametrize("rows,cols,valid", [(5, 64, 64), (7, 96, 81), (3, 4096, 4096), (2, 9216, 9216), (4, 32, 1)])
def test_softmax_rows(out_dtype, rows, cols, valid):
    from ldm import ops
    s = (seeded((rows, cols), 3) * 9.0).cuda()
    scale = 512 ** -0.5
    y = ops.softmax_rows(s, valid, scale, out_dtype)
    ref = torch.zeros(rows, cols, dtype=torch.float64)
    ref[:, :valid] = torch.softmax(s[:, :valid].double().cpu() * scale, dim=-1)
    assert y.dtype == out_dtype
    assert rel_l2(y, ref) < (1e-6 if out_dtype == torch.float32 else 4e-3)
    assert float(y[:, valid:].float().abs().max() if valid < cols else 0.0) == 0.0


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv1x1_nchw(dtype):
    from ldm import ops
    x = seeded((3, 4, 5, 7), 1).to(dtype)
    w, b = seeded((4, 4, 1, 1), 2), seeded((4,), 3)
    y = ops.conv1x1_nchw(x.cuda(), w.cuda(), b.cuda())
    ref = F.conv2d(x.double(), w.double(), b.double())
    assert y.dtype == torch.float32 and rel_l2(y, ref) < 1e-6


@pytest.mark.parametrize("dtype", DTYPES)
def test_image_to_uint8_is_torchs_expression_bit_for_bit(dtype):
    from ldm import ops
    x = (seeded((2, 3, 64, 64), 5) * 0.9).to(dtype).cuda()
    x.view(-1)[:6] = torch.tensor([-1.0, 1.0, -3.0, 3.0, 0.0, 0.999], dtype=dtype, device="cuda")
    got = ops.image_to_uint8(x)
    ref = vref.to_uint8(x)                      # train_latent_diffusion.py:600-601 evaluated by torch in the same dtype
    assert got.dtype == torch.uint8 and torch.equal(got, ref)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_in_with_512_output_channels(dtype):
    from ldm import ops
    x, w, b = seeded((2, 4, 9, 6), 1), seeded((512, 4, 3, 3), 2) / 6.0, seeded((512,), 3)
    y = ops.conv_in(x.cuda(), w.cuda(), b.cuda(), dtype)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1)
    assert y.shape == (2, 9, 6, 512) and rel_l2(y, ref) < (1e-6 if dtype == torch.float32 else 4e-3)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,hw", [(128, 3, (256, 256)), (320, 4, (260, 253)), (64, 1, (300, 300))])
def test_conv_out_on_large_images(dtype, cin, cout, hw):
    """>= 65536 pixels take the 16-lanes-per-pixel kernel (filter in LDS); 320 channels = 40 chunks, not a multiple of 16."""
    from ldm import ops
    x = seeded((1, *hw, cin), 1).to(dtype)
    w, b = seeded((cout, cin, 3, 3), 2) / (3.0 * cin ** 0.5), seeded((cout,), 3)
    y = ops.conv_out(x.cuda(), w.cuda(), b.cuda(), dtype)
    ref = F.conv2d(x.double().permute(0, 3, 1, 2), w.double(), b.double(), padding=1)
    assert y.shape == (1, cout, *hw) and rel_l2(y, ref) < (1e-6 if dtype == torch.float32 else 4e-3)


# ------------------------------------------------------------------------------------------- blocks
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("C,hw", [(64, 8), (64, 6), (512, 16)])
def test_mid_block_attention(dtype, C, hw):
    """hw=6: 36 positions, not a multiple of the GEMM's K granularity (padded, masked columns)."""
    from ldm import vae
    p = "decoder.mid_block.attentions.0"
    shapes = {k: v for k, v in vref.param_shapes(vref.VAEDecoderConfig(block_out_channels=(C, C))).items() if k.startswith(p)}
    sd = vref.fill_state_dict(shapes, seed=21)
    att = vae._Attention(C, 32)
    att.load_state_dict({k[len(p) + 1:]: v for k, v in sd.items()})
    holder = vae._HipModule()
    holder.add_module("decoder", torch.nn.Module())
    holder.decoder.add_module("mid_block", torch.nn.Module())
    holder.decoder.mid_block.add_module("attentions", torch.nn.ModuleList([att]))
    holder = holder.cuda()
    x = seeded((2, C, hw, hw), 22)
    ref = vref.attention(_f64(sd), p, x.double(), 32)
    P = holder._packed_for(dtype)
    with torch.no_grad():
        y = vae.run_attention(P, p, att, x.cuda().to(dtype).permute(0, 2, 3, 1).contiguous()).permute(0, 3, 1, 2)
    assert rel_l2(y, ref) < TOL[dtype]


def test_mid_block_attention_at_the_64x64_latent_extent():
    """The shape the real decoder runs: one head of 512 channels over 4096 positions (bf16)."""
    from ldm import vae
    C, hw, p = 512, 64, "decoder.mid_block.attentions.0"
    shapes = {k: v for k, v in vref.param_shapes(vref.VAEDecoderConfig()).items() if k.startswith(p)}
    sd = vref.fill_state_dict(shapes, seed=23)
    att = vae._Attention(C, 32)
    att.load_state_dict({k[len(p) + 1:]: v for k, v in sd.items()})
    holder = vae._HipModule()
    holder.add_module("decoder", torch.nn.Module())
    holder.decoder.add_module("mid_block", torch.nn.Module())
    holder.decoder.mid_block.add_module("attentions", torch.nn.ModuleList([att]))
    holder = holder.cuda()
    x = seeded((1, C, hw, hw), 24)
    ref = vref.attention({k: v.cuda() for k, v in sd.items()}, p, x.cuda(), 32)      # the oracle, evaluated by torch in fp32
    P = holder._packed_for(torch.bfloat16)
    with torch.no_grad():
        y = vae.run_attention(P, p, att, x.cuda().bfloat16().permute(0, 2, 3, 1).contiguous()).permute(0, 3, 1, 2)
    assert rel_l2(y, ref) < 2e-2


# ------------------------------------------------------------------------------------------ whole decoder
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hw", [(8, 8), (6, 10)])
def test_small_decoder_matches_the_float64_oracle(dtype, hw):
    m, cfg, sd = _vae(SMALL)
    z = seeded((2, 4, *hw), 31)
    ref = vref.decode(_f64(sd), cfg, z.double())
    img = m.decode(z.cuda().to(dtype), return_dict=False)[0]
    assert img.dtype == dtype and img.shape == (2, 3, 8 * hw[0], 8 * hw[1])
    assert rel_l2(img, ref) < TOL[dtype]
    # the reference's post-processing (:600-601): uint8 images within one level of the oracle's
    u8 = __import__("ldm.vae", fromlist=["to_uint8"]).to_uint8(img)
    lvl = (u8.cpu().int() - vref.to_uint8(ref.float()).int()).abs()
    assert int(lvl.max()) <= (1 if dtype == torch.float32 else 8) and float((lvl > 0).float().mean()) < (0.02 if dtype == torch.float32 else 0.6)


def test_bf16_autocast_of_float32_latents_is_the_bf16_path():
    m, cfg, sd = _vae(SMALL)
    z = seeded((1, 4, 8, 8), 32).cuda()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        a = m.decode(z).sample
    ref = vref.decode(_f64(sd), cfg, z.double().cpu())
    assert a.dtype == torch.bfloat16 and rel_l2(a, ref) < 2e-2       # float32 latents are not rounded on the way in
    assert m.decode(z).sample.dtype == torch.float32


def test_micro_batches_do_not_change_the_images():
    m, cfg, sd = _vae(SMALL)
    z = seeded((3, 4, 8, 8), 33).cuda().bfloat16()
    whole = m.decode(z).sample
    parts = m.decode(z, micro_batch=1).sample
    assert rel_l2(parts, whole) < 2e-3       # tile choice may differ with the row count; same arithmetic otherwise
    assert m.decode(z[:0]).sample.shape == (0, 3, 64, 64)


def test_full_size_decoder_bf16_against_the_oracle_evaluated_in_fp32_on_the_gpu():
    """SD-1.x decoder (49.5 M parameters), one 32x32 latent -> 256x256 image: the reference's default image size
    (VAE.md:87-156)."""
    m, cfg, sd = _vae({}, seed=41)
    z = seeded((1, 4, 32, 32), 42).cuda()
    ref = vref.decode({k: v.cuda() for k, v in sd.items()}, cfg, z)
    img = m.decode(z.bfloat16()).sample
    assert img.shape == (1, 3, 256, 256)
    assert rel_l2(img, ref) < 2e-2


def test_rejects_cpu_tensors_and_wrong_channel_counts():
    m, cfg, sd = _vae(SMALL)
    with pytest.raises(RuntimeError):
        m.decode(torch.zeros(1, 4, 8, 8))
    with pytest.raises(ValueError):
        m.decode(torch.zeros(1, 3, 8, 8, device="cuda"))
    with pytest.raises(NotImplementedError):
        m.encode(torch.zeros(1, 3, 64, 64, device="cuda"))


def test_decode_images_is_the_reference_tail_of_generate():
    """ldm/sampling.py::decode_images = :596-601: divide by the scaling factor, decode (bf16 autocast), convert to uint8."""
    from ldm import sampling
    from ldm.vae import to_uint8
    m, cfg, sd = _vae(SMALL)
    lat = seeded((3, 4, 8, 8), 51).cuda()
    got = sampling.decode_images(m, lat, micro_batch=2)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        img = m.decode(lat / m.config.scaling_factor, return_dict=False)[0]
    want = to_uint8(img)
    assert got.dtype == torch.uint8 and got.shape == (3, 3, 64, 64)
    lvl = (got.int() - want.int()).abs()
    assert int(lvl.max()) <= 2            # micro-batches of 2 vs one batch of 3: same arithmetic, tile choice may differ
    ref = vref.to_uint8(vref.decode(_f64(sd), cfg, (lat.cpu() / cfg.scaling_factor).double()).float())
    assert float((got.cpu().int() - ref.int()).abs().float().mean()) < 1.5
