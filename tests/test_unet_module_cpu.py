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


def test_public_names_of_the_reference_modules():
    """Every public class / function of the reference's ldm/unet.py (:159-448, :451-790) exists under the same name with
    the same constructor arguments; the stage classes own exactly the parameters of the matching sub-tree of the whole
    net's state_dict, and the UNet is built out of them."""
    import inspect
    from ldm import unet as U
    cfg = uref.UNetConfig(block_out_channels=[32, 64, 64, 64])
    shapes = uref.param_shapes(cfg)

    def subtree(prefix):
        return {k[len(prefix) + 1:]: v for k, v in shapes.items() if k.startswith(prefix + ".")}

    def check(mod, prefix):
        sd, want = mod.state_dict(), subtree(prefix)
        assert list(sd.keys()) == list(want.keys()), prefix
        assert all(tuple(sd[k].shape) == tuple(want[k]) for k in want), prefix

    kw = dict(num_attention_heads=8, cross_attention_dim=768)
    check(U.CrossAttnDownBlock(32, 32, 128, num_layers=2, **kw), "down_blocks.0")
    check(U.CrossAttnDownBlock(32, 64, 128, num_layers=2, **kw), "down_blocks.1")
    check(U.DownBlock(64, 64, 128, num_layers=2), "down_blocks.3")
    check(U.CrossAttnBlock(64, 64, 128, num_layers=1, **kw), "mid_block")
    check(U.UpBlock(64, 64, 64, 128, num_layers=3), "up_blocks.0")
    check(U.CrossAttnUpBlock(64, 64, 32, 128, num_layers=3, **kw), "up_blocks.2")
    last = U.CrossAttnUpBlock(64, 32, 32, 128, num_layers=3, add_upsample=False, **kw)
    check(last, "up_blocks.3")
    assert last.upsample is None
    net = U.UNet(block_out_channels=[32, 64, 64, 64])
    assert [type(b).__name__ for b in net.down_blocks] == ["CrossAttnDownBlock"] * 3 + ["DownBlock"]
    assert type(net.mid_block).__name__ == "CrossAttnBlock"
    assert [type(b).__name__ for b in net.up_blocks] == ["UpBlock"] + ["CrossAttnUpBlock"] * 3
    # constructor signatures: names, order and defaults of the reference (unet.py:178-191, :241-249, :277-286, :322-337, :391-404)
    def sig(c):
        return [(p.name, p.default) for p in list(inspect.signature(c.__init__).parameters.values())[1:]]
    E = inspect.Parameter.empty
    assert sig(U.DownBlock) == [("in_channels", E), ("out_channels", E), ("temb_channels", E), ("dropout", 0.0), ("num_layers", 1),
                                ("norm_num_groups", 32)]
    assert sig(U.UpBlock) == [("in_channels", E), ("out_channels", E), ("final_skip_channels", E), ("temb_channels", E),
                              ("dropout", 0.0), ("num_layers", 1), ("norm_num_groups", 32)]
    attn_tail = [("transformer_layers_per_block", 1), ("norm_num_groups", 32), ("num_attention_heads", 8),
                 ("cross_attention_dim", 1280), ("feed_forward_mult", 4), ("activation_fn", "geglu")]
    assert sig(U.CrossAttnDownBlock) == [("in_channels", E), ("out_channels", E), ("temb_channels", E), ("dropout", 0.0),
                                         ("num_layers", 2)] + attn_tail
    assert sig(U.CrossAttnBlock) == [("in_channels", E), ("out_channels", E), ("temb_channels", E), ("dropout", 0.0),
                                     ("num_layers", 1)] + attn_tail
    assert sig(U.CrossAttnUpBlock) == [("in_channels", E), ("out_channels", E), ("final_skip_channels", E), ("temb_channels", E),
                                       ("dropout", 0.0), ("num_layers", 1)] + attn_tail + [("add_upsample", True)]
    assert list(inspect.signature(U.get_time_embed).parameters) == ["sample", "timestep", "num_channels"]
    for name in ("UNet", "ResnetBlock", "Attention", "TransformerBlock", "TransformerModel", "GEGLU", "SwiGLU"):
        assert hasattr(U, name)
    # .to(memory_format=...) keeps the checkpoint layout (train_latent_diffusion.py:876 must not re-stride the filters)
    net2 = net.to(memory_format=torch.channels_last)
    assert net2 is net and net.conv_in.weight.is_contiguous()
    assert net.up_blocks[0].upsample.weight.is_contiguous()


def test_dropout_is_accepted_and_refused_only_where_it_would_act():
    """nn.Dropout(p) sits in ResnetBlock.layers, Attention.output_proj and the feed-forward (unet.py:707, :636-639, :579).  It is
    the identity in eval mode, so a model built with dropout > 0 must construct, load and sample; in training mode the
    reference draws masks, for which there is no kernel: that call is refused instead of silently ignored (checked before
    the device check, so it shows on a CPU tensor)."""
    from ldm import unet as U
    x = torch.zeros(1, 32, 8, 8)
    for mod, args in ((U.ResnetBlock(32, 32, 64, 8, dropout=0.1), (x, torch.zeros(1, 64))),
                      (U.Attention(32, 4, 8, dropout=0.1), (torch.zeros(1, 16, 32),)),
                      (U.TransformerBlock(32, 4, 48, dropout=0.1), (torch.zeros(1, 16, 32), torch.zeros(1, 5, 48))),
                      (U.UNet(block_out_channels=[32, 64, 64, 64], dropout=0.2), (torch.zeros(1, 4, 8, 8), 3, torch.zeros(1, 77, 768)))):
        assert mod._active_dropout() > 0
        with pytest.raises(NotImplementedError, match="eval mode"):
            mod(*args)                                    # module.training is True after construction
        mod.eval()
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            mod(*args)                                    # past the dropout check: only the device check is left
    plain = U.ResnetBlock(32, 32, 64, 8)
    assert plain._active_dropout() == 0.0
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        plain(x, torch.zeros(1, 64))
