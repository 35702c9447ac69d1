"""CPU: the drop-in UNet module owns exactly the reference's parameters (state_dict key contract)
and refuses to compute without a GPU."""
import pytest
import torch

from oracle import unet_ref as uref


@pytest.mark.parametrize("kwargs", [dict(block_out_channels=[32, 64, 64, 64]),
                                    dict(block_out_channels=[32, 64], layers_per_block=1, activation_fn="swiglu")])
def test_state_dict_matches_reference_contract(kwargs):
    from ldm.unet import UNet
    net = UNet(**kwargs)
    shapes = uref.param_shapes(uref.UNetConfig(**kwargs))
    sd = net.state_dict()
    assert list(sd.keys()) == list(shapes.keys())
    for k, v in sd.items():
        assert tuple(v.shape) == tuple(shapes[k]), k
    net.load_state_dict(uref.fill_state_dict(shapes, seed=1))
    assert net.block_out_channels == kwargs["block_out_channels"]


def test_default_config_on_meta_device():
    from ldm.unet import UNet
    with torch.device("meta"):
        net = UNet()
    shapes = uref.param_shapes(uref.UNetConfig())
    sd = net.state_dict()
    assert list(sd.keys()) == list(shapes.keys()) and len(sd) == 654
    assert sum(v.numel() for v in sd.values()) == 859520964
    assert all(tuple(sd[k].shape) == tuple(shapes[k]) for k in shapes)


def test_constructor_errors_and_cpu_refusal():
    from ldm.unet import UNet, TransformerBlock
    with pytest.raises(AssertionError):
        UNet(block_out_channels=[32])
    with pytest.raises(ValueError):
        TransformerBlock(64, 8, 768, activation_fn="relu")
    net = UNet(block_out_channels=[32, 64])
    with torch.no_grad(), pytest.raises(RuntimeError):
        net(torch.zeros(1, 4, 8, 8), 3, torch.zeros(1, 77, 768))
