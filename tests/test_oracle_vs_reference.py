"""Pin the oracle against the reference executed live (only where /root/reference exists)."""
import pytest
import torch

import helpers as H
from oracle import medt_oracle as O
from oracle import ref_loader

pytestmark = pytest.mark.skipif(not ref_loader.available(), reason="reference checkout not present")


@pytest.mark.parametrize("name,S,N", [("gatedaxialunet", 64, 2), ("axialunet", 64, 1), ("MedT", 128, 1), ("logo", 128, 1)])
def test_model_forward_backward_fp64(name, S, N):
    torch.manual_seed(0)
    ref = ref_loader.factory(name)(img_size=S, imgchan=3)
    sd = O.randomize_state(ref.state_dict(), 3)
    ref.load_state_dict(sd)
    ref = ref.double()
    for p in ref.parameters():
        p.requires_grad_(True)
    x, y = H.seeded_input(4, N, 3, S)
    for training in (True, False):
        ref.train(training)
        st = O.clone_state(ref.state_dict(), torch.float64, requires_grad=True)
        out_ref = ref(x.double())
        out = O.forward(name, x.double(), st, training)
        assert H.rel_err(out, out_ref) < 1e-9
        if training:
            loss_ref = ref_loader.load_metrics().LogNLLLoss()(out_ref, y)
            loss = O.log_nll_loss(out, y)
            assert abs(loss.item() - loss_ref.item()) < 1e-11
            loss_ref.backward()
            loss.backward()
            gmax = max(p.grad.abs().max().item() for p in ref.parameters() if p.grad is not None)
            for k, p in ref.named_parameters():
                if p.grad is None:
                    continue
                assert (st[k].grad - p.grad).abs().max().item() < 1e-8 * gmax, k
            for k, b in ref.state_dict().items():
                if "running" in k or "num_batches" in k:
                    assert H.rel_err(st[k].double(), b.double()) < 1e-9, k


def test_reference_gray_and_manifest_agree():
    m = ref_loader.factory("MedT")(img_size=128, imgchan=1)
    ent = H.manifest()["MedT/128/1"]["state"]
    assert [k for k, _, _ in ent] == list(m.state_dict().keys())


@pytest.mark.parametrize("cls_name,mode", [("AxialAttention_gated_sig", "sigmoid"), ("AxialAttention_gated_data", "data")])
@pytest.mark.parametrize("width,stride", [(False, 1), (True, 2)])
def test_gate_variants_fp64(cls_name, mode, width, stride):
    """The experimental gate flavours of the reference's model_codes.py (:215-313, :316-443) against the oracle's
    gate_mode, forward + every gradient, training mode, executed live in fp64."""
    mc = ref_loader.load_model_codes()
    torch.manual_seed(5)
    layer = getattr(mc, cls_name)(32, 32, groups=8, kernel_size=12, stride=stride, width=width).double()
    for p in layer.parameters():
        if p.dim() == 0:
            p.data.uniform_(-1, 1)
        p.requires_grad_(True)
    layer.train()
    x = torch.randn(3, 32, 12, 12, dtype=torch.float64, requires_grad=True)
    out_ref = layer(x)
    w = torch.randn_like(out_ref)
    (out_ref * w).sum().backward()
    st = O.clone_state({"L." + k: v for k, v in layer.state_dict().items()}, torch.float64, requires_grad=True)
    xo = x.detach().clone().requires_grad_(True)
    out = O.axial_attention(xo, st, "L", width, stride, True, gate_mode=mode)
    (out * w).sum().backward()
    assert H.rel_err(out, out_ref) < 1e-10
    assert H.rel_err(xo.grad, x.grad) < 1e-9
    gmax = max(p.grad.abs().max().item() for p in layer.parameters() if p.grad is not None)
    for k, p in layer.named_parameters():
        assert (st["L." + k].grad - p.grad).abs().max().item() < 1e-9 * gmax, k


def test_gated_sig_module_surface_matches_reference():
    import lib as droplib  # noqa: F401
    from lib.models import model_codes
    mc = ref_loader.load_model_codes()
    torch.manual_seed(9)
    a = mc.AxialAttention_gated_sig(32, 32, groups=8, kernel_size=16, stride=2, width=True)
    torch.manual_seed(9)
    b = model_codes.AxialAttention_gated_sig(32, 32, groups=8, kernel_size=16, stride=2, width=True)
    sa, sb = a.state_dict(), b.state_dict()
    assert list(sa.keys()) == list(sb.keys())
    for k in sa:
        assert torch.equal(sa[k], sb[k]), k                # same registration order -> same RNG stream -> same init
    assert [(k, p.requires_grad) for k, p in a.named_parameters()] == [(k, p.requires_grad) for k, p in b.named_parameters()]
    torch.manual_seed(10)
    c = mc.AxialAttention_gated_data(32, 32, groups=8, kernel_size=16, stride=1, width=False)
    torch.manual_seed(10)
    d = model_codes.AxialAttention_gated_data(32, 32, groups=8, kernel_size=16, stride=1, width=False)
    sc, sd = c.state_dict(), d.state_dict()
    assert list(sc.keys()) == list(sd.keys())              # fcn1 / fcn2 sit between bn_output and relative, as in the reference
    for k in sc:
        assert torch.equal(sc[k], sd[k]), k
    blk = model_codes.AxialBlock_gated_data(32, 16, kernel_size=16)
    ref_blk = mc.AxialBlock_gated_data(32, 16, kernel_size=16)
    assert list(blk.state_dict().keys()) == list(ref_blk.state_dict().keys())
