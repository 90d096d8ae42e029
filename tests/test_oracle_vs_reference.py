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
