"""CPU checks of the VAE-decoder leg: the state-dict contract and the oracle's restatement.

PARITY UNPINNED (see oracle/vae_ref.py): the only external anchors available offline are the reference's own drawing of the
layer stack (VAE.md:119-157) and the published size of the SD-1.x AutoencoderKL decoder.
"""
import pytest
import torch

from conftest import seeded
from oracle import vae_ref as vref


def test_default_decoder_has_the_published_parameter_count():
    shapes = vref.param_shapes(vref.VAEDecoderConfig())
    n = lambda pre: sum(int(torch.Size(s).numel()) for k, s in shapes.items() if k.startswith(pre))
    assert n("decoder.") == 49_490_179          # AutoencoderKL(SD-1.x).decoder
    assert n("post_quant_conv.") == 20


def test_module_state_dict_is_the_oracles_key_and_shape_contract():
    from ldm.vae import AutoencoderKL
    for kw in ({}, dict(block_out_channels=(32, 64, 64, 64)), dict(block_out_channels=(32, 64), layers_per_block=1)):
        with torch.device("meta"):
            m = AutoencoderKL(**kw)
        got = {k: tuple(v.shape) for k, v in m.state_dict().items()}
        assert got == vref.param_shapes(vref.VAEDecoderConfig(**kw))


def test_full_checkpoints_load_with_the_encoder_half_ignored():
    from ldm.vae import AutoencoderKL
    kw = dict(block_out_channels=(32, 64))
    cfg = vref.VAEDecoderConfig(**kw)
    sd = vref.fill_state_dict(vref.param_shapes(cfg), seed=3)
    sd["encoder.conv_in.weight"] = torch.zeros(32, 3, 3, 3)
    sd["quant_conv.weight"] = torch.zeros(8, 8, 1, 1)
    m = AutoencoderKL(**kw)
    m.load_state_dict(sd)
    assert torch.equal(m.decoder.conv_in.weight, sd["decoder.conv_in.weight"])
    assert m.config.scaling_factor == 0.18215
    with pytest.raises(RuntimeError):
        m.decode(torch.zeros(1, 4, 4, 4))       # no CPU fallback


def test_oracle_shapes_and_upsampling_factor():
    cfg = vref.VAEDecoderConfig(block_out_channels=(32, 64, 64))
    sd = vref.fill_state_dict(vref.param_shapes(cfg), seed=5)
    z = seeded((2, 4, 4, 6), 6)
    img = vref.decode(sd, cfg, z)
    assert img.shape == (2, 3, 16, 24)
    # samples are independent (GroupNorm and attention are per sample): what micro-batched decoding relies on
    one = vref.decode(sd, cfg, z[1:])
    assert torch.allclose(one, img[1:], atol=1e-5, rtol=1e-5)
    u8 = vref.to_uint8(torch.tensor([-1.0, 1.0, 0.0, 3.0, -3.0]))
    assert u8.tolist() == [0, 255, 127, 255, 0]
