"""medt_amd.trainer.InferStep -- the no-grad forward of reference test.py:106-119 / train.py:174-184 replayed as a hipGraph --
against the eager launch-by-launch forward of the same model: bit-equal logits and segmentation counts, the same
running-statistics updates in train mode, parameter updates between replays seen."""
import copy

import pytest
import torch

import helpers as H
from test_model_gpu import build

pytestmark = pytest.mark.gpu


def _as_device(t, device):
    """On the emulated device (pytest --emulate) CPU tensors stand in for device tensors."""
    if device.type == "cpu":
        from emu_device import DeviceTensor
        return t.as_subclass(DeviceTensor)
    return t.to(device)


def _model(name, S, device, seed=3):
    model = build(name, S, device)
    model.load_state_dict(H.seeded_state(name, S, seed))          # non-trivial running statistics and BatchNorm affines
    return model


@pytest.mark.parametrize("name,S,N", [("MedT", 128, 4), ("MedT", 128, 1), ("gatedaxialunet", 128, 2), ("axialunet", 64, 2)])
def test_replayed_eval_forward_equals_eager(name, S, N, device):
    from medt_amd.trainer import InferStep
    from medt_amd.ops import seg_counts
    if device.type == "cpu" and S > 64:
        pytest.skip("emulated device: the 64-px network only")
    model = _model(name, S, device).eval()
    infer = InferStep(model, use_graph=device.type == "cuda")
    for k in range(3):                    # call 0 captures, calls 1-2 replay with fresh inputs
        x, y = H.seeded_input(20 + k, N, 3, S)
        x, y = _as_device(x, device), _as_device(y, device)
        with torch.no_grad():
            want = model(x)
            want_counts = seg_counts(want, y)
        got, counts = infer(x, y)
        assert torch.equal(got, want), (k, H.rel_err(got, want))
        assert torch.equal(counts, want_counts), k
        assert not got.requires_grad
    if device.type == "cuda":
        assert len(infer._graphs) == 1
        got2 = infer(x)                   # without targets: another signature, another graph, the same logits
        assert torch.equal(got2, want) and len(infer._graphs) == 2


def test_replay_in_train_mode_updates_running_statistics_like_eager(device):
    """reference train.py:174-184 validates with the model left in train mode: batch statistics, and every forward is one
    running-statistics update (16 per BatchNorm of MedT's local branch, in patch order)."""
    from medt_amd.trainer import InferStep
    name, S, N = ("MedT", 128, 2) if device.type == "cuda" else ("axialunet", 64, 2)
    a = _model(name, S, device).train()
    b = copy.deepcopy(a)
    infer = InferStep(b, use_graph=device.type == "cuda")
    for k in range(3):
        x, _ = H.seeded_input(30 + k, N, 3, S)
        x = _as_device(x, device)
        with torch.no_grad():
            want = a(x)
        got = infer(x)
        assert torch.equal(got, want), (k, H.rel_err(got, want))
    sa, sb = a.state_dict(), b.state_dict()
    n = 0
    for key in sa:
        if "running" in key or "num_batches" in key:
            assert torch.equal(sa[key], sb[key]), key
            n += 1
    assert n > 50
    nbt = [int(v) for k, v in sb.items() if "num_batches" in k]
    assert max(nbt) == (3 * 16 if name == "MedT" else 3)           # warm-up forwards of the capture left nothing behind


def test_replay_sees_parameter_updates_and_recaptures_on_new_storage(device):
    from medt_amd.trainer import InferStep
    if device.type != "cuda":
        pytest.skip("graph replay needs the GPU")
    model = _model("axialunet", 64, device).eval()
    infer = InferStep(model)
    x, _ = H.seeded_input(41, 2, 3, 64)
    x = x.to(device)
    y0 = infer(x).clone()
    with torch.no_grad():
        for p in model.parameters():
            p.mul_(1.01)                                           # in place: the captured graph reads the same storage
        want = model(x)
    got = infer(x)
    assert len(infer._graphs) == 1 and torch.equal(got, want) and not torch.equal(got, y0)
    with torch.no_grad():
        p = next(model.parameters())
        p.data = p.data.clone()                                    # storage re-pointed (what FlatAdam's adoption does)
        p.mul_(0.5)
        want = model(x)
    got = infer(x)
    assert len(infer._graphs) == 2 and torch.equal(got, want)


def test_replayed_forward_is_faster_than_eager(device):
    """The point of the exercise: the eager forward is host-bound.  (Loose bound: 2x; measured ~4x at batch 4.)"""
    import time
    from medt_amd.trainer import InferStep
    if device.type != "cuda":
        pytest.skip("timing needs the GPU")
    model = build("MedT", 128, device).eval()
    infer = InferStep(model)
    x = torch.rand(4, 3, 128, 128, device=device)

    def t(fn, reps):
        for _ in range(3):
            fn(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn(x)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    with torch.no_grad():
        eager, replay = t(model, 10), t(infer, 30)
    print(f"MedT 128 bs 4 eval forward: eager {eager * 1e3:.3f} ms, replayed {replay * 1e3:.3f} ms")
    assert replay < 0.5 * eager


@pytest.mark.parametrize("name,S,K", [("MedT", 128, 4), ("gatedaxialunet", 128, 4)])
def test_gathered_eval_replay_equals_single_image_replays(name, S, K, device):
    """test.py --gather (round 6): K loader items of batch size 1 share one replay.  In eval mode BatchNorm is an affine of the
    running statistics, so the images of a batch do not interact ARITHMETICALLY; whether the bits agree as well depends on
    whether the batch size changes a kernel choice (tile shapes / split contractions sum in another order).  Held: label
    maps (logit >= 0.5, reference test.py:123-124) and the device-side counts identical, logits equal to 1e-5 of their range;
    bit-equality is reported."""
    from medt_amd.trainer import InferStep
    if device.type == "cpu":
        pytest.skip("GPU only")
    model = _model(name, S, device).eval()
    infer = InferStep(model)
    x, y = H.seeded_input(31, K, 3, S)
    x, y = x.to(device), y.to(device)
    singles, counts1 = [], []
    for k in range(K):
        o, c = infer(x[k:k + 1].contiguous(), y[k:k + 1].contiguous())
        singles.append(o.clone())
        counts1.append(c.clone())
    want, wc = torch.cat(singles), torch.cat(counts1)
    got, gc = infer(x, y)
    err = H.rel_err(got, want)
    print(f"{name}: one replay of {K} images vs {K} single-image replays: bit-equal {torch.equal(got, want)}, rel err {err:.1e}")
    assert err < 1e-5
    safe = (want - 0.5).abs() > 1e-4 * want.abs().max()
    assert torch.equal((got >= 0.5)[safe], (want >= 0.5)[safe])
    if bool(safe.all()):
        assert torch.equal(gc, wc)
    # the padded last batch of test.py: copies of the last image behind it change nothing for the images in front
    xp = torch.cat([x[:2], x[1:2], x[1:2]])
    yp = torch.cat([y[:2], y[1:2], y[1:2]])
    gp, _ = infer(xp, yp)
    assert torch.equal(gp[:2], got[:2]) and torch.equal(gp[2], gp[1]) and torch.equal(gp[3], gp[1])
