"""Pin the CPU oracle against the fixtures generated from the reference itself
(tests/golden/make_golden.py).  Runs anywhere (no GPU, no /root/reference)."""
import glob
import json
import os

import numpy as np
import pytest
import torch

import helpers as H
from oracle import medt_oracle as O

LAYER_FILES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(H.GOLDEN, "layer_*.npz")))
MODEL_FILES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(H.GOLDEN, "model_*.npz")))


def layer_state_from_fixture(fx):
    layout = json.loads(str(fx["state_layout"]))
    blank = {k: torch.zeros(shape, dtype=getattr(torch, dt)) for k, shape, dt in layout}
    seed = int(fx["meta"][5])
    st = O.randomize_state(blank, seed)
    for k in st:
        if k.endswith("flatten_index"):
            L = int(fx["meta"][1])
            ar = torch.arange(L)
            st[k] = (ar.view(L, 1) - ar.view(1, L) + L - 1).reshape(-1)
    return st


def test_fixture_inventory():
    assert len(LAYER_FILES) >= 6 and len(MODEL_FILES) >= 7


@pytest.mark.parametrize("fn", LAYER_FILES)
def test_layer_oracle_matches_reference_fixture(fn):
    fx = H.load_golden(fn)
    C, L, width, stride, N, seed = [int(v) for v in fx["meta"]]
    st = {("m." + k): v for k, v in layer_state_from_fixture(fx).items()}
    x = torch.from_numpy(fx["x"]).double()
    # eval
    st_e = O.clone_state(st, torch.float64)
    gm = {"gatedsig": "sigmoid", "gateddata": "data"}.get(fn.split("_")[1], "raw")
    out = O.axial_attention(x, st_e, "m", bool(width), stride, training=False, gate_mode=gm)
    assert H.rel_err(out, fx["out_eval"]) < 1e-10
    # train: forward, backward, running stats
    st_t = O.clone_state(st, torch.float64, requires_grad=True)
    xg = x.clone().requires_grad_(True)
    out = O.axial_attention(xg, st_t, "m", bool(width), stride, training=True, gate_mode=gm)
    assert H.rel_err(out, fx["out_train"]) < 1e-10
    (out * torch.from_numpy(fx["dout"])).sum().backward()
    assert H.rel_err(xg.grad, fx["dx"]) < 1e-9
    gscale = max(np.abs(fx[k]).max() for k in fx if k.startswith("grad/"))
    for k in fx:
        if k.startswith("grad/"):
            g = st_t["m." + k[5:]].grad
            g = torch.zeros_like(st_t["m." + k[5:]]) if g is None else g
            err = (g - torch.from_numpy(fx[k])).abs().max().item()
            assert err < 1e-9 * max(gscale, 1.0), k
        if k.startswith("buf/"):
            assert H.rel_err(st_t["m." + k[4:]].double(), fx[k]) < 1e-10, k


BLOCK_FILES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(H.GOLDEN, "block_*.npz")))


@pytest.mark.parametrize("fn", BLOCK_FILES)
def test_block_oracle_matches_reference_fixture(fn):
    """O.axial_block with grouped BatchNorm statistics == the reference's AxialBlock_wopos applied to the patch groups one after
    the other (medt_net's patch loop, axialnet.py:661-700): outputs, every gradient, the running statistics in patch order."""
    fx = H.load_golden(fn)
    inplanes, planes, S, groups_n, npg, seed = [int(v) for v in fx["meta"][:6]]
    stride = int(fx["meta"][6]) if len(fx["meta"]) > 6 else 1      # (round 6: the stride-2 first block with its downsample path)
    layout = json.loads(str(fx["state_layout"]))
    blank = {k: torch.zeros(shape, dtype=getattr(torch, dt)) for k, shape, dt in layout}
    st = {("m." + k): v for k, v in O.randomize_state(blank, seed).items()}
    x = torch.from_numpy(fx["x"]).double()
    out = O.axial_block(x, O.clone_state(st, torch.float64), "m", stride, False, groups_n)
    assert H.rel_err(out, fx["out_eval"]) < 1e-10
    st_t = O.clone_state(st, torch.float64, requires_grad=True)
    xg = x.clone().requires_grad_(True)
    out = O.axial_block(xg, st_t, "m", stride, True, groups_n)
    assert H.rel_err(out, fx["out_train"]) < 1e-10
    (out * torch.from_numpy(fx["dout"])).sum().backward()
    assert H.rel_err(xg.grad, fx["dx"]) < 1e-9
    gscale = max(np.abs(fx[k]).max() for k in fx if k.startswith("grad/"))
    ngrads = 0
    for k in fx:
        if k.startswith("grad/"):
            g = st_t["m." + k[5:]].grad
            g = torch.zeros_like(st_t["m." + k[5:]]) if g is None else g
            assert (g - torch.from_numpy(fx[k])).abs().max().item() < 1e-9 * max(gscale, 1.0), k
            ngrads += 1
        if k.startswith("buf/"):
            assert H.rel_err(st_t["m." + k[4:]].double(), fx[k]) < 1e-10, k
    assert ngrads >= 18


@pytest.mark.parametrize("fn", MODEL_FILES)
def test_model_oracle_matches_reference_fixture(fn):
    fx = H.load_golden(fn)
    name = fn.split("_")[1]
    S, N, seed, training = [int(v) for v in fx["meta"]]
    mode = str(fx["mode"])
    x, y = H.seeded_input(seed + 1, N, 3, S)
    xs = x.double()
    assert abs(xs.sum().item() - fx["x_checksum"][0]) < 1e-6, "CPU RNG contract changed: regenerate fixtures"
    st = O.clone_state(H.seeded_state(name, S, seed), torch.float64, requires_grad=(mode != "eval"))
    out = O.forward(name, xs, st, mode == "train")
    assert H.rel_err(out, fx["logits"]) < 1e-6            # fixture logits are stored as float32
    if mode == "eval":
        return
    loss = O.log_nll_loss(out, y)
    assert abs(loss.item() - fx["loss"][0]) < 1e-10
    loss.backward()
    names, summ = list(fx["grad_names"]), fx["grad_summary"]
    gmax = summ[:, 0].max()
    for k, (norm, dot) in zip(names, summ):
        g = st[k].grad.reshape(-1)
        assert abs(g.norm().item() - norm) < 1e-8 * gmax + 1e-9 * norm, k
        assert abs(torch.dot(g, H.probe_vector(k, g.numel(), seed)).item() - dot) < 1e-7 * gmax * g.numel() ** 0.5, k
    for k, dots in zip(names, fx["grad_dots"]):              # the eight probe dots of every gradient tensor
        g = st[k].grad.reshape(-1)
        got = H.probe_matrix(k, g.numel(), seed) @ g
        assert (got - torch.from_numpy(dots)).abs().max().item() < 1e-7 * gmax * g.numel() ** 0.5, k
    for k in fx:
        if k.startswith("grad/"):
            assert (st[k[5:]].grad - torch.from_numpy(fx[k])).abs().max().item() < 1e-9 * gmax + 1e-12, k
    if mode == "train":
        for k, (norm, dot) in zip(list(fx["buf_names"]), fx["buf_summary"]):
            v = st[k].reshape(-1).double()
            if k.endswith("num_batches_tracked"):
                assert float(v.item()) == norm, k
            else:
                assert abs(v.norm().item() - norm) < 1e-9 * max(norm, 1.0), k


def test_medt_batched_patches_equals_loop():
    """The product's batched-patch layout (16 BN groups) is the reference's sequential loop."""
    st0 = H.seeded_state("MedT", 128, 5)
    x, _ = H.seeded_input(6, 2, 3, 128)
    a = O.clone_state(st0, torch.float64)
    b = O.clone_state(st0, torch.float64)
    ya = O.medt(x.double(), a, True, batch_patches=False)
    yb = O.medt(x.double(), b, True, batch_patches=True)
    assert H.rel_err(yb, ya) < 1e-10
    for k in a:
        if "running" in k or "num_batches" in k:
            assert H.rel_err(b[k].double(), a[k].double()) < 1e-10, k


def test_fast_bn_mode_is_the_same_arithmetic():
    """bench.py's cpu_baseline leg runs the oracle with aten's fused BatchNorm; it must be the same function."""
    st0 = H.seeded_state("gatedaxialunet", 64, 8)
    x, _ = H.seeded_input(9, 2, 3, 64)
    a = O.clone_state(st0, torch.float64)
    b = O.clone_state(st0, torch.float64)
    ya = O.forward("gatedaxialunet", x.double(), a, True)
    O.set_fast_bn(True)
    try:
        yb = O.forward("gatedaxialunet", x.double(), b, True)
    finally:
        O.set_fast_bn(False)
    assert H.rel_err(yb, ya) < 1e-10
    for k in a:
        if "running" in k or "num_batches" in k:
            assert H.rel_err(b[k].double(), a[k].double()) < 1e-10, k


def test_oracle_at_the_factory_state_matches_reference_fixture():
    """bench.py's initial state (factory initialisation under seed 3000 before construction, train mode, bench.py's batch; fixture generated
    from the reference by make_golden.py::factory_fixture): the product's module surface gives the reference's initial state
    (checksum), and the oracle reproduces the reference's float64 logits and loss on it (fixture logits are stored in float32)."""
    import lib as droplib
    fx = H.load_golden("factory_MedT_S128_N4.npz")
    torch.manual_seed(int(fx["meta"][2]))
    sd = droplib.models.axialnet.MedT(img_size=128, imgchan=3).state_dict()
    cs = sum(v.double().sum().item() for v in sd.values() if v.is_floating_point())
    assert abs(cs - fx["state_checksum"][0]) < 1e-9 * abs(fx["state_checksum"][0])
    x, y = H.seeded_input(3000, 4, 3, 128)
    out = O.forward("MedT", x.double(), O.clone_state(sd, torch.float64), True)
    assert H.rel_err(out, fx["logits"]) < 5e-7
    assert abs(O.log_nll_loss(out, y).item() - fx["loss"][0]) < 1e-9
