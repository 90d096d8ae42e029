"""Drop-in boundary: the mirrored lib.models.axialnet has the reference's state_dict layout."""
import pytest
import torch

import helpers as H
from oracle import ref_loader


def build(name, S, chan=3):
    import lib as droplib
    f = {"axialunet": droplib.models.axialunet, "gatedaxialunet": droplib.models.axialnet.gated,
         "MedT": droplib.models.axialnet.MedT, "logo": droplib.models.axialnet.logo}[name]
    return f(img_size=S, imgchan=chan)


@pytest.mark.parametrize("key", sorted(H.manifest().keys()))
def test_state_dict_layout(key):
    name, S, chan = key.split("/")
    m = build(name, int(S), int(chan))
    ent = H.manifest()[key]
    got = [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in m.state_dict().items()]
    assert got == ent["state"]
    assert [[k, bool(p.requires_grad)] for k, p in m.named_parameters()] == ent["params"]


def test_public_names():
    import lib as droplib
    ax = droplib.models.axialnet
    for n in ("AxialAttention", "AxialAttention_dynamic", "AxialAttention_wopos", "AxialBlock", "AxialBlock_dynamic",
              "AxialBlock_wopos", "ResAxialAttentionUNet", "medt_net", "axialunet", "gated", "MedT", "logo",
              "qkv_transform", "conv1x1"):
        assert hasattr(ax, n), n
    assert hasattr(droplib.models, "axialunet")          # train.py:96 uses lib.models.axialunet
    a = ax.AxialAttention_dynamic(16, 16, groups=8, kernel_size=8, stride=2, width=True)
    for attr in ("in_planes", "out_planes", "groups", "group_planes", "kernel_size", "stride", "bias", "width"):
        assert hasattr(a, attr)
    assert ax.AxialBlock_dynamic.expansion == 2


@pytest.mark.skipif(not ref_loader.available(), reason="reference checkout not present")
@pytest.mark.parametrize("name", ["gatedaxialunet", "MedT"])
def test_same_seed_same_init_as_reference(name):
    torch.manual_seed(1234)
    ref = ref_loader.factory(name)(img_size=128, imgchan=3)
    torch.manual_seed(1234)
    mine = build(name, 128)
    a, b = ref.state_dict(), mine.state_dict()
    assert list(a.keys()) == list(b.keys())
    for k in a:
        assert torch.equal(a[k], b[k]), k
    # a reference checkpoint loads strictly
    mine.load_state_dict(ref.state_dict(), strict=True)
