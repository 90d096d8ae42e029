"""Whole-network parity on the GPU: HIP path vs the reference-generated fixtures and the CPU oracle."""
import glob
import os

import numpy as np
import pytest
import torch

import helpers as H
from oracle import medt_oracle as O

pytestmark = pytest.mark.gpu
MODEL_FILES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(H.GOLDEN, "model_*.npz")))
TOL = 1e-3
KNOISE = 2.0        # training-mode gradients: product error <= KNOISE x the reference's own measured fp32 noise
                    # (max over EIGHT float32 runs of the reference per tensor -- make_golden.py; observed on the MI355X:
                    #  max ratio 0.65 - 0.91 over the four training fixtures, median 0.07 - 0.20)


def build(name, S, device, chan=3):
    import lib as droplib
    f = {"axialunet": droplib.models.axialunet, "gatedaxialunet": droplib.models.axialnet.gated,
         "MedT": droplib.models.axialnet.MedT, "logo": droplib.models.axialnet.logo}[name]
    return f(img_size=S, imgchan=chan).to(device)


def check_gradient_summaries(fx, params, seed, mode, knoise, tol, noise_floor_rel=None):
    """Every gradient tensor of the model against the fixture's summaries: the L2 norm and EIGHT seeded probe dots
    (make_golden.py; rounds 1-3 stored one, which left a norm-preserving error of a big tensor to a single 3-sigma test).
    evalgrad: `tol` relative.  train: knoise x the reference's OWN float32 noise on this tensor (max over eight fp32 runs
    of the reference), never less than `tol` (or `noise_floor_rel`, bf16 storage) of the tensor's scale.  A probe has
    unit-variance entries: an error vector of norm e moves a dot product by ~N(0, e^2), so each dot is held to 3 sigma of
    knoise x the noise NORM (or the measured dot noise, if larger).  -> (violations, ratios, gmax)."""
    names, summ, noise = list(fx["grad_names"]), fx["grad_summary"], fx["grad_noise"]
    dots, dots_noise = fx["grad_dots"], fx["grad_dots_noise"]
    gmax = summ[:, 0].max()
    bad, ratios = [], []
    for k, (norm, _), (nz, _), want_d, nzd in zip(names, summ, noise, dots, dots_noise):
        g = params[k].grad.double().cpu().reshape(-1)
        scale = max(norm, 1e-3 * gmax)
        d = (H.probe_matrix(k, g.numel(), seed) @ g).numpy()
        derr = float(np.abs(d - want_d).max())
        if mode == "evalgrad":
            tol_n = tol_d = tol * scale
            tol_d *= 5
        else:
            floor = max(nz, (noise_floor_rel if noise_floor_rel is not None else tol) * scale)
            tol_n, tol_d = knoise * floor, 3 * knoise * max(floor, nzd)
            ratios.append((max(abs(g.norm().item() - norm) / floor, derr / (3 * max(floor, nzd))), k))
        if abs(g.norm().item() - norm) > tol_n:
            bad.append((k, "norm", g.norm().item(), norm))
        if derr > tol_d:
            bad.append((k, "dots", derr, tol_d))
    return bad, ratios, gmax


@pytest.mark.parametrize("fn", MODEL_FILES)
def test_model_vs_reference_fixture(fn, device):
    """Tolerances.  eval / evalgrad (running statistics): 1e-3 relative, everything.
    train (batch statistics): logits 1e-3 or 1.5x the reference's own fp32-vs-fp64 discrepancy, whichever
    is larger; every gradient tensor within KNOISE x the reference's own fp32 noise on THAT tensor (the
    maximum over eight fp32 runs of the reference -- other summation orders, inputs / parameters moved by one
    unit in the last place -- stored per tensor by make_golden.py) -- the whole-network training-mode backward is ill-conditioned in fp32 for the
    reference itself (DESIGN.md 'parity floor').  The ratio product-error / reference-noise is printed.
    The exact backward wiring is pinned by the evalgrad fixtures and by the layer-level tests at 1e-3."""
    fx = H.load_golden(fn)
    name = fn.split("_")[1]
    S, N, seed, _ = [int(v) for v in fx["meta"]]
    mode = str(fx["mode"])
    model = build(name, S, device)
    model.load_state_dict(H.seeded_state(name, S, seed))
    for p in model.parameters():
        p.requires_grad_(True)
    model.train(mode == "train")
    x, y = H.seeded_input(seed + 1, N, 3, S)
    if mode == "eval":
        with torch.no_grad():
            out = model(x.to(device))
    else:
        out = model(x.to(device))
    want = torch.from_numpy(fx["logits"])
    err = H.rel_err(out, want)
    ltol = max(TOL, 1.5 * float(fx["logits_noise"][0])) if mode == "train" else TOL
    assert err < ltol, err
    if mode != "train":
        # label maps (reference thresholds logits at 0.5, train.py:144-145) and argmax: bit-exact away from ties
        o = out.detach().double().cpu()
        safe = (want.double() - 0.5).abs() > 1e-3 * want.abs().max()
        assert torch.equal((o >= 0.5)[safe], (want >= 0.5)[safe])
        margin = (want[:, 1] - want[:, 0]).abs() > 1e-3 * want.abs().max()
        assert torch.equal(o.argmax(1)[margin], want.argmax(1)[margin])
        print(f"{fn}: label map (logit >= 0.5) bit-exact on {int(safe.sum())} of {safe.numel()} values "
              f"({int((~safe).sum())} within 1e-3 of the threshold excluded); argmax bit-exact on {int(margin.sum())} of "
              f"{margin.numel()} pixels ({int((~margin).sum())} near-ties excluded)")
    if mode == "eval":
        return
    loss = torch.nn.functional.cross_entropy(out, y.to(device))
    assert abs(loss.item() - fx["loss"][0]) < (1e-4 if mode == "evalgrad" else 1e-3)
    loss.backward()
    torch.cuda.synchronize()
    params = dict(model.named_parameters())
    bad, ratios, gmax = check_gradient_summaries(fx, params, seed, mode, KNOISE, TOL)
    if ratios:
        r = np.sort(np.array([v for v, _ in ratios]))
        print(f"{fn}: product error / reference fp32 noise per gradient tensor: median {np.median(r):.2f}, "
              f"90% {r[int(0.9 * len(r))]:.2f}, max {r[-1]:.2f} (bound {KNOISE}); largest: "
              + ", ".join(f"{k} {v:.2f}" for v, k in sorted(ratios, reverse=True)[:5]))
    assert not bad, bad[:8]
    for k in fx:
        if k.startswith("grad/"):
            w = torch.from_numpy(fx[k])
            scale = max(w.abs().max().item(), 1e-3 * gmax)
            tol_abs = TOL * scale if mode == "evalgrad" else KNOISE * max(float(fx["gradnoise/" + k[5:]][0]), TOL * scale)
            assert (params[k[5:]].grad.double().cpu() - w).abs().max().item() < tol_abs, k
    if mode == "train":
        sd = model.state_dict()
        for k, (norm, dot), nz in zip(list(fx["buf_names"]), fx["buf_summary"], fx["buf_noise"]):
            v = sd[k].double().cpu().reshape(-1)
            if k.endswith("num_batches_tracked"):
                assert float(v.item()) == norm, k
            else:
                tol_abs = max(TOL * max(norm, 1e-3), 4 * nz)          # reference's own fp32 noise on this buffer
                assert abs(v.norm().item() - norm) < tol_abs, k
                assert abs(torch.dot(v, H.probe_vector(k, v.numel(), seed)).item() - dot) < 5 * tol_abs, k


def test_gated_bs8_train_bf16_storage_vs_reference_conditioning(device):
    """BASELINE.json configs[1] in TRAINING mode: gatedaxialunet, 128 px, batch 8, bf16 activation storage, against the
    reference's float64 results for that batch (fixture model_gatedaxialunet_S128_N8_train; test_model_vs_reference_fixture
    holds the fp32 path to it at the reference's fp32 noise).

    There is no tolerance this mode can be held to, FOR THE REFERENCE ITSELF: its float64 train-mode logits move by 42 % of
    their range when the INPUT IMAGE alone is rounded once to bfloat16 precision (tests/golden/sensitivity_*.json, generated
    from the reference by make_golden.py: 1.5e-4 for one fp32 rounding, 3.2e-2 for 2^-16, 0.42 for 2^-9 -- batch-statistic
    BatchNorm through ~100 layers amplifies ~2000x until it saturates).  bf16 storage inside the network is at least that
    perturbation, so the product is judged against the reference's own response to it: logits within 2x that change, finite
    loss / gradients, loss within 0.15 of the reference's.  bf16 parity proper (3e-2, every gradient) is held in
    running-statistics mode by test_baseline_batch_sizes_evalgrad_vs_oracle and at layer level in training mode by
    test_layer_bf16_storage_vs_oracle."""
    import json
    import medt_amd
    fn = "model_gatedaxialunet_S128_N8_train.npz"
    fx = H.load_golden(fn)
    with open(os.path.join(H.GOLDEN, "sensitivity_gatedaxialunet_S128_N8.json")) as f:
        sens = json.load(f)
    S, N, seed, _ = [int(v) for v in fx["meta"]]
    assert (sens["S"], sens["N"], sens["seed"]) == (S, N, seed)
    model = build("gatedaxialunet", S, device)
    model.load_state_dict(H.seeded_state("gatedaxialunet", S, seed))
    for p in model.parameters():
        p.requires_grad_(True)
    model.train()
    x, y = H.seeded_input(seed + 1, N, 3, S)
    medt_amd.set_activation_dtype(torch.bfloat16)
    try:
        out = model(x.to(device))
        loss = torch.nn.functional.cross_entropy(out, y.to(device))
        loss.backward()
        torch.cuda.synchronize()
    finally:
        medt_amd.set_activation_dtype(torch.float32)
    err = H.rel_err(out, torch.from_numpy(fx["logits"]))
    ref_move = sens["logits_rel_change_for_input_rounding"]["bf16"]
    print(f"{fn} with bf16 storage, TRAIN mode: logits rel err {err:.3f} vs the reference's own fp64 response to ONE bf16 "
          f"rounding of its input {ref_move:.3f} (fp32 rounding: {sens['logits_rel_change_for_input_rounding']['f32']:.1e}); "
          f"loss {loss.item():.4f} vs {fx['loss'][0]:.4f}")
    assert err < 2.0 * ref_move, (err, ref_move)
    assert abs(loss.item() - fx["loss"][0]) < 0.15
    for k, p in model.named_parameters():
        assert p.grad is None or torch.isfinite(p.grad).all(), k


def test_gated_bs8_train_bf16_storage_at_the_factory_state(device):
    """Round-5 verdict, weak #2: the 'no tolerance exists' finding above was made on randomize_state weights; at bench.py's
    FACTORY state the same network's fp32 noise is 2.6e-3 - 4.2e-3, so bf16 storage might be holdable to ~5e-2 there.
    Measured from the reference (tests/golden/sensitivity_factory_gatedaxialunet_S128_N8.json, make_golden.py, round 6): it is NOT
    better conditioned -- the reference's float64 train-mode logits at the factory state move by 5.2e-4 for one fp32 rounding
    of the input, 9.0e-2 for 2^-16 and **0.41 for one bf16 rounding** (randomize_state: 1.5e-4 / 3.2e-2 / 0.42): the
    amplification saturates at either state.  The finding is about training-mode gatedaxialunet, not about the weights.
    The product with bf16 storage is therefore judged, here too, against the reference's own response: logits within 2 x that
    move, the loss within 0.15, finite gradients; the figures (max and MEDIAN deviation) are printed."""
    import json
    import medt_amd
    name, S, N = "gatedaxialunet", 128, 8
    fx = H.load_golden(f"factory_{name}_S{S}_N{N}.npz")
    with open(os.path.join(H.GOLDEN, "sensitivity_factory_gatedaxialunet_S128_N8.json")) as f:
        sens = json.load(f)
    seed = int(fx["meta"][2])
    assert (sens["S"], sens["N"], sens["seed"], sens["state"]) == (S, N, seed, "factory")
    torch.manual_seed(seed)
    model = build(name, S, device)
    x, y = H.seeded_input(seed, N, 3, S)
    model.train()
    want = torch.from_numpy(fx["logits"]).double()
    res = {}
    for dt in (torch.float32, torch.bfloat16):
        torch.manual_seed(seed)
        model = build(name, S, device)
        model.train()
        medt_amd.set_activation_dtype(dt)
        try:
            out = model(x.to(device))
            loss = torch.nn.functional.cross_entropy(out, y.to(device))
            loss.backward()
            torch.cuda.synchronize()
        finally:
            medt_amd.set_activation_dtype(torch.float32)
        d = (out.detach().double().cpu() - want).abs() / want.abs().max()
        res[dt] = (d.max().item(), d.median().item(), loss.item())
        for k, p in model.named_parameters():
            assert p.grad is None or torch.isfinite(p.grad).all(), k
    ref_move = sens["logits_rel_change_for_input_rounding"]["bf16"]
    print(f"factory state {name} bs {N} TRAIN: fp32 storage max {res[torch.float32][0]:.2e} / median {res[torch.float32][1]:.2e}; "
          f"bf16 storage max {res[torch.bfloat16][0]:.2e} / median {res[torch.bfloat16][1]:.2e}; the reference's own float64 response to one "
          f"bf16 rounding of its input: {ref_move:.2f} (fp32 rounding {sens['logits_rel_change_for_input_rounding']['f32']:.1e}); "
          f"loss {res[torch.bfloat16][2]:.4f} / {res[torch.float32][2]:.4f} vs {fx['loss'][0]:.4f}")
    assert res[torch.bfloat16][0] < 2.0 * ref_move
    assert abs(res[torch.bfloat16][2] - fx["loss"][0]) < 0.15


def test_gated_evalgrad_vs_oracle_full(device):
    """Running-statistics mode, every gradient tensor compared in full against the live fp64 oracle."""
    name, S, N = "gatedaxialunet", 64, 2
    model = build(name, S, device)
    st = O.randomize_state({k: v.cpu() for k, v in model.state_dict().items()}, 21)
    model.load_state_dict(st)
    for p in model.parameters():
        p.requires_grad_(True)
    model.eval()
    x, y = H.seeded_input(22, N, 3, S)
    out = model(x.to(device))
    torch.nn.functional.cross_entropy(out, y.to(device)).backward()
    ost = O.clone_state(st, torch.float64, requires_grad=True)
    oout = O.forward(name, x.double(), ost, False)
    O.log_nll_loss(oout, y).backward()
    assert H.rel_err(out, oout) < 1e-4
    gmax = max(v.grad.abs().max().item() for v in ost.values() if v.grad is not None)
    for k, p in model.named_parameters():
        g = ost[k].grad
        if g is None:
            assert p.grad is None or p.grad.abs().max().item() == 0, k
            continue
        scale = max(g.abs().max().item(), 1e-3 * gmax)
        assert (p.grad.double().cpu() - g).abs().max().item() / scale < TOL, k


@pytest.mark.parametrize("name,S,N,bf16", [("MedT", 128, 4, False), ("gatedaxialunet", 128, 8, False),
                                           ("gatedaxialunet", 128, 8, True)],
                         ids=["MedT128-bs4", "gated128-bs8", "gated128-bs8-bf16"])
def test_baseline_batch_sizes_evalgrad_vs_oracle(name, S, N, bf16, device):
    """BASELINE.json's own batch sizes (configs[2]: MedT bs=4; configs[1]: gatedaxialunet bs=8, fp32 and bf16 activation
    storage) in running-statistics mode against the live fp64 oracle: logits, label maps (with the number of near-tie
    pixels that are excluded stated), loss, every gradient.  fp32: 1e-3; bf16 storage: 3e-2 (2^-9 rounding per stored
    element through 16 attention layers)."""
    import medt_amd
    tol = 3e-2 if bf16 else TOL
    model = build(name, S, device)
    st = O.randomize_state({k: v.cpu() for k, v in model.state_dict().items()}, 61)
    model.load_state_dict(st)
    for p in model.parameters():
        p.requires_grad_(True)
    model.eval()
    x, y = H.seeded_input(62, N, 3, S)
    medt_amd.set_activation_dtype(torch.bfloat16 if bf16 else torch.float32)
    try:
        out = model(x.to(device))
        loss = torch.nn.functional.cross_entropy(out, y.to(device))
        loss.backward()
        torch.cuda.synchronize()
    finally:
        medt_amd.set_activation_dtype(torch.float32)
    ost = O.clone_state(st, torch.float64, requires_grad=True)
    oout = O.forward(name, x.double(), ost, False)
    oloss = O.log_nll_loss(oout, y)
    oloss.backward()
    err = H.rel_err(out, oout)
    assert err < tol, err
    assert abs(loss.item() - oloss.item()) < tol * max(1.0, abs(oloss.item()))
    o, w = out.detach().double().cpu(), oout.detach()
    margin = max(1e-3, 5 * err) * w.abs().max()                  # near-tie band: 5x the measured logit error
    safe = (w - 0.5).abs() > margin
    excluded = int((~safe).sum())
    assert torch.equal((o >= 0.5)[safe], (w >= 0.5)[safe])
    am = (w[:, 1] - w[:, 0]).abs() > margin
    assert torch.equal(o.argmax(1)[am], w.argmax(1)[am])
    print(f"{name} bs={N} bf16={bf16}: logits rel err {err:.2e}; label map bit-exact on {int(safe.sum())} of {safe.numel()} "
          f"values ({excluded} within {float(margin):.1e} of the 0.5 threshold excluded), argmax on {int(am.sum())} of {am.numel()}")
    gmax = max(v.grad.abs().max().item() for v in ost.values() if v.grad is not None)
    worst = 0.0
    for k, p in model.named_parameters():
        g = ost[k].grad
        if g is None:
            assert p.grad is None or p.grad.abs().max().item() == 0, k
            continue
        scale = max(g.abs().max().item(), 1e-3 * gmax)
        e = (p.grad.double().cpu() - g).abs().max().item() / scale
        worst = max(worst, e)
        assert e < tol, (k, e)
    print(f"  worst gradient rel err {worst:.2e}")


def test_training_trajectory_vs_oracle(device):
    """Three optimisation steps of the reference's loop (train.py:140,156-161: forward, LogNLLLoss, backward, Adam with
    lr 1e-3 / weight_decay 1e-5) through the hipGraph-replayed product step against the fp64 oracle driving
    oracle.adam_step: the loss of every step -- measured in units of what float32 oracle runs of the same loop deviate by
    -- and the BatchNorm running statistics after the last one.  (Weights are not compared element-wise: Adam turns the
    rounding noise of a near-zero gradient into a +-lr step.)"""
    import medt_amd
    from medt_amd.optim import FlatAdam
    from medt_amd.trainer import TrainStep
    name, S, N, STEPS = "gatedaxialunet", 64, 2, 3
    torch.manual_seed(80)                                    # the factory's own initialisation, reproducibly
    model = build(name, S, device)
    model.train()
    st = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    x, y = H.seeded_input(81, N, 3, S)
    opt = FlatAdam(list(model.parameters()), lr=1e-3, weight_decay=1e-5)
    step = TrainStep(model, opt, medt_amd.cross_entropy, use_graph=True, warmup=2)
    losses = [step(x.to(device), y.to(device)).item() for _ in range(STEPS)]
    train_keys = [k for k, p in model.named_parameters() if p.requires_grad]

    def oracle_run(dtype, ulp_seed=None):
        """The same loop on the CPU oracle; parameters that receive a gradient are updated, the frozen gates are not.
        ulp_seed: the initial weights and the input moved by one float32 ulp in random directions first."""
        ost, xin = O.clone_state(st, dtype), x.to(dtype)
        if ulp_seed is not None:
            g = torch.Generator().manual_seed(ulp_seed)
            nudge = lambda t: torch.nextafter(t, torch.where(torch.rand(t.shape, generator=g) < 0.5, -1.0, 1.0) * float("inf"))
            ost = {k: (nudge(v) if (v.is_floating_point() and k in train_keys) else v) for k, v in ost.items()}
            xin = nudge(xin)
        mom, out_losses = {}, []
        for t in range(1, STEPS + 1):
            leaf = {k: (v.clone().requires_grad_(True) if k in train_keys else v) for k, v in ost.items()}
            loss = O.log_nll_loss(O.forward(name, xin, leaf, True), y)
            loss.backward()
            out_losses.append(loss.item())
            for k in ost:
                if k in train_keys and leaf[k].grad is not None:
                    m, v = mom.get(k, (torch.zeros_like(ost[k]), torch.zeros_like(ost[k])))
                    pnew, m, v = O.adam_step(ost[k], leaf[k].grad, m, v, t)
                    ost[k], mom[k] = pnew.detach(), (m, v)
                else:
                    ost[k] = leaf[k].detach()                    # buffers (running statistics) as updated by the forward
        return out_losses, ost

    want, ost = oracle_run(torch.float64)
    # Adam turns the rounding noise of every near-zero gradient into a +-lr step, so from the second step on the loss of
    # ANY float32 run leaves the float64 trajectory by far more than float32 rounding.  The yardstick is therefore the
    # reference's own arithmetic: three float32 oracle trajectories (as is, and from 1-ulp-perturbed starts).
    f32_runs = [oracle_run(torch.float32, sd_)[0] for sd_ in (None, 811, 812)]
    for t, (a, b) in enumerate(zip(losses, want)):
        noise = max(abs(r[t] - b) for r in f32_runs)
        assert abs(a - b) <= 2.0 * noise + 2e-6 * abs(b), (t, losses, want, f32_runs)
        assert abs(a - b) <= 1e-2 * abs(b), (t, losses, want)          # and never more than a percent
    print(f"trajectory: product {losses} | f64 oracle {want} | f32 oracle runs {f32_runs}")
    sd = model.state_dict()
    # (shallow layers only: the deep ones normalise over 32-value populations whose statistics follow every rounding)
    for k in ("bn1.running_mean", "bn2.running_var", "layer1.0.hight_block.bn_similarity.running_var",
              "layer1.0.hight_block.bn_qkv.running_mean"):
        # (weights have moved by up to 3 lr-sized Adam steps, in noise-driven directions where the gradient is ~0:
        #  3.2e-2 observed on bn1.running_mean in 1 of ~10 runs; a wrong momentum or count is a > 10 % error)
        assert H.rel_err(sd[k], ost[k]) < 8e-2, k
    assert int(sd["bn1.num_batches_tracked"].item()) == STEPS


def test_graphed_train_step_equals_eager(device):
    """The hipGraph-replayed step (trainer.TrainStep) IS the eager step, bit for bit: the warm-up steps before capture are
    rolled back (weights, Adam state, BatchNorm running statistics, num_batches_tracked), so capture + N replays perform
    exactly N updates.  MedT's position-encoded layers all take the single-sweep backward (no float atomics) and every
    reduction of the step has a fixed order, so losses, weights, Adam moments, running statistics and counters after one
    and after four steps are EQUAL.  (The eager run spends its FlatAdam adoption step -- gradients through autograd's `.grad`,
    immediate instead of recorded weight-gradient launches, i.e. another fp32 summation order -- in a rolled-back warm-up
    too, like the captured run does.)"""
    import medt_amd
    from medt_amd.optim import FlatAdam
    from medt_amd.trainer import TrainStep
    name, S, N = "MedT", 128, 2
    st = H.seeded_state(name, S, 33)
    x, y = H.seeded_input(34, N, 3, S)
    x, y = x.to(device), y.to(device)
    results = []
    for use_graph in (False, True):
        model = build(name, S, device)
        model.load_state_dict(st)
        model.train()
        opt = FlatAdam(list(model.parameters()), lr=1e-3, weight_decay=1e-5)
        step = TrainStep(model, opt, medt_amd.cross_entropy, use_graph=use_graph, warmup=2)
        if not use_graph:
            snap = step._snapshot()
            step._eager(x, y)
            step._restore(snap)
        losses = [step(x, y).item()]
        first = {k: v.detach().clone() for k, v in model.state_dict().items()}
        losses += [step(x, y).item() for _ in range(3)]
        torch.cuda.synchronize()
        g = opt.groups[0]
        results.append((losses, first, {k: v.detach().clone() for k, v in model.state_dict().items()},
                        [g.state.clone(), g.exp_avg.clone(), g.exp_avg_sq.clone()]))
    (l0, f0, s0, o0), (l1, f1, s1, o1) = results
    assert l0 == l1, (l0, l1)
    moved = 0
    for k in f0:
        assert torch.equal(f0[k], f1[k]), ("after the first step", k)
        assert torch.equal(s0[k], s1[k]), ("after four steps", k)
        if f0[k].is_floating_point() and "running" not in k and k in st:
            moved += int((f0[k].cpu() != st[k]).any())
    assert moved > 200                                              # the first call did update the weights
    assert int(s0["layer1_p.0.bn1.num_batches_tracked"].item()) == 4 * 16       # 4 steps x 16 patches, not 4 + warm-up
    for a, b in zip(o0, o1):
        assert torch.equal(a, b)
    assert float(o0[0][0]) == 4.0                                   # Adam's step counter


def test_deferred_grouped_launches_match_immediate(device):
    """medt_queue_*: recording the off-chain work (weight / bias gradients, slab reductions, statistics bookkeeping of the
    fused small-layer kernels) and issuing it as grouped launches gives the same numbers as the immediate launches --
    same kernel bodies; the weight gradients are chunked differently, so they agree to fp32 rounding -- for every gradient,
    running statistic and counter.  (Gradients are only
    recorded when they land in FlatAdam's persistent slots, so both runs go through an adopted optimizer.)"""
    import medt_amd
    from medt_amd.defer import StepQueue
    from medt_amd.optim import FlatAdam
    name, S, N = "MedT", 128, 2
    st = H.seeded_state(name, S, 71)
    x, y = H.seeded_input(72, N, 3, S)
    x, y = x.to(device), y.to(device)
    runs = []
    for deferred in (False, True):
        model = build(name, S, device)
        model.load_state_dict(st)
        model.train()
        opt = FlatAdam(list(model.parameters()), lr=0.0)
        medt_amd.cross_entropy(model(x), y).backward()      # adoption step (immediate launches in both runs)
        opt.pack_gradients()
        model.load_state_dict(st)                           # running statistics / counters back to the start
        opt.zero_grad()
        if deferred:
            q = StepQueue()
            with q.active():
                loss = medt_amd.cross_entropy(model(x), y)
                assert q.pending() + q.issued > 30          # the forward recorded bookkeeping jobs (each branch issues its
                q.flush()                                   # own where its forward ends: net.medt_forward, MEDT_EARLY_FIN)
                before = q.issued
                loss.backward()
                assert q.pending() + q.issued - before > 100   # the backward recorded the weight-gradient jobs
            assert q.pending() == 0
        else:
            loss = medt_amd.cross_entropy(model(x), y)
            loss.backward()
        opt.pack_gradients()
        torch.cuda.synchronize()
        runs.append((loss.item(), {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None},
                     {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}))
    (l0, g0, b0), (l1, g1, b1) = runs
    assert l0 == l1
    assert g0.keys() == g1.keys() and len(g0) > 200
    for k in g0:
        if k.endswith("relative") or (g0[k].dim() > 1):     # LDS float atomics (run-to-run order) / the grouped weight-gradient
            assert H.rel_err(g1[k], g0[k]) < 1e-5, k        # launch chunks the positions differently (fp32 summation order)
        else:
            assert torch.equal(g0[k], g1[k]), k              # BatchNorm / bias / gate gradients: same kernel bodies
    for k in b0:
        assert torch.equal(b0[k], b1[k]), k


def test_failed_step_leaves_no_recorded_jobs(device):
    """A step that raises (class index outside [0, K), what F.cross_entropy raises on) must not leave recorded jobs in
    the library's queues: they point at tensors the failed step releases.  The forward's bookkeeping is flushed (running
    statistics advance exactly as in the reference, whose forward also completed), nothing stays pending, and the next
    good step gives bit-identical running statistics / counters / loss to a model that ran the same forward + step without
    any failure; an exception in the middle of a pass drops its recorded jobs (medt_queue_discard)."""
    import medt_amd
    from medt_amd import MedtError
    from medt_amd.defer import StepQueue
    from medt_amd.optim import FlatAdam
    from medt_amd.trainer import TrainStep
    name, S, N = "MedT", 128, 2
    st = H.seeded_state(name, S, 91)
    x, y = H.seeded_input(92, N, 3, S)
    x, y = x.to(device), y.to(device)
    y_bad = y.clone()
    y_bad[0, 3, 5] = 7
    out = []
    for fail in (True, False):
        model = build(name, S, device)
        model.load_state_dict(st)
        model.train()
        opt = FlatAdam(list(model.parameters()), lr=0.0)
        step = TrainStep(model, opt, medt_amd.cross_entropy, use_graph=False)
        if fail:
            with pytest.raises((MedtError, RuntimeError, IndexError)):
                step(x, y_bad)
            assert step._queue.pending() == 0
        else:
            with torch.no_grad():
                model(x)                                    # the failed step's forward, without the failure
        loss = step(x, y).item()
        torch.cuda.synchronize()
        out.append((loss, {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}))
    (l0, b0), (l1, b1) = out
    assert l0 == l1
    for k in b0:
        assert torch.equal(b0[k], b1[k]), k
    assert int(b0["layer1_p.0.bn1.num_batches_tracked"].item()) == 32
    # an exception in the middle of a pass: recorded jobs are dropped, not left for the next flush
    # (a single-branch network: MedT's two branches issue their forward bookkeeping themselves where each one ends)
    model = build("gatedaxialunet", 64, device)
    model.train()
    x2, y2 = H.seeded_input(93, N, 3, 64)
    q = StepQueue()
    with pytest.raises(ZeroDivisionError):
        with q.active():
            medt_amd.cross_entropy(model(x2.to(device)), y2.to(device))
            assert q.pending() > 10
            raise ZeroDivisionError
    assert q.pending() == 0


def test_flat_adam_slots_are_written_directly(device):
    """After adoption the backward kernels write parameter gradients straight into FlatAdam's flat bucket:
    `.grad` is a view of it, nothing is returned through autograd, and the bucket equals the unbound gradients."""
    import medt_amd
    from medt_amd.optim import FlatAdam
    name, S, N = "gatedaxialunet", 64, 2
    st = H.seeded_state(name, S, 51)
    x, y = H.seeded_input(52, N, 3, S)
    x, y = x.to(device), y.to(device)
    model = build(name, S, device)
    model.load_state_dict(st)
    model.eval()                                            # running statistics: repeatable
    for p in model.parameters():
        p.requires_grad_(True)                              # gates included: four adjacent 0-d slots per layer
    medt_amd.cross_entropy(model(x), y).backward()
    want = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    opt = FlatAdam(list(model.parameters()), lr=0.0)
    opt.pack_gradients()                                    # adopts everything that has a gradient
    opt.zero_grad()
    medt_amd.cross_entropy(model(x), y).backward()
    for k, p in model.named_parameters():                   # nothing came back through autograd
        assert p.grad is None, k
    opt.pack_gradients()
    assert len(opt.groups) == 1
    flat = opt.groups[0].flat_g
    lo, hi = flat.data_ptr(), flat.data_ptr() + flat.numel() * 4
    for k, p in model.named_parameters():
        if k in want:
            assert lo <= p.grad.data_ptr() < hi, k
            # (bit-equal except where LDS float atomics decide the summation order: the relative tables)
            assert H.rel_err(p.grad, want[k]) < 1e-5, k
    # outside the zero_grad() .. pack_gradients() window the slots are not touched: a second backward accumulates into
    # `.grad` the way plain autograd does (here: doubles it), instead of silently overwriting or double counting
    medt_amd.cross_entropy(model(x), y).backward()
    for k, p in model.named_parameters():
        if k in want:
            assert H.rel_err(p.grad, 2 * want[k]) < 1e-5, k
    # torch.autograd.grad works too and leaves the buckets alone
    before = flat.clone()
    w0 = next(model.parameters())
    (g0,) = torch.autograd.grad(medt_amd.cross_entropy(model(x), y), [w0])
    assert torch.equal(flat, before)
    assert H.rel_err(g0, want[next(iter(want))]) < 1e-5
    # freezing a parameter that has been trained is refused (torch.optim.Adam would skip it; the flat kernel cannot)
    opt.zero_grad()
    w0.requires_grad_(False)
    medt_amd.cross_entropy(model(x), y).backward()
    with pytest.raises(medt_amd.MedtError):
        opt.pack_gradients()


def test_medt_256_train_vs_oracle(device):
    """BASELINE.json config 5 geometry (MedT, 256 px: L = 128 attention, patches only cover the top-left 128x128,
    SURVEY.md Q1), training mode, against the live oracle at the state of the reference fixture model_MedT_S256_N2_train.npz
    (round 6; test_model_vs_reference_fixture holds every gradient, running statistic and num_batches_tracked of that fixture).
    Logits: max(1e-3, 1.5 x the reference's own fp32-vs-fp64 deviation on this state) -- the fixture's `logits_noise`, 6.9e-4 over
    eight float32 runs of the reference, i.e. a bound of 1.04e-3 (rounds 3-5 used a flat 3e-3 here)."""
    name, S, N = "MedT", 256, 2
    fx = H.load_golden("model_MedT_S256_N2_train.npz")
    seed = int(fx["meta"][2])
    model = build(name, S, device)
    st = H.seeded_state(name, S, seed)
    model.load_state_dict(st)
    model.train()
    x, y = H.seeded_input(seed + 1, N, 3, S)
    out = model(x.to(device))
    loss = torch.nn.functional.cross_entropy(out, y.to(device))
    loss.backward()
    torch.cuda.synchronize()
    ost = O.clone_state(st, torch.float64)
    oout = O.forward(name, x.double(), ost, True)
    assert H.rel_err(oout, torch.from_numpy(fx["logits"])) < 1e-6            # the live oracle IS the reference's float64 run
    bound = max(1e-3, 1.5 * float(fx["logits_noise"][0]))
    err = H.rel_err(out, oout)
    print(f"MedT 256 px bs 2 train: logits rel err {err:.2e} (bound {bound:.2e} = max(1e-3, 1.5 x the reference's fp32 noise))")
    assert err < bound, (err, bound)
    assert abs(loss.item() - O.log_nll_loss(oout, y).item()) < 1e-3
    sd = model.state_dict()
    for k in ("bn1.running_mean", "layer1.0.hight_block.bn_similarity.running_var", "layer4_p.0.bn2.running_mean",
              "layer2_p.1.width_block.bn_output.running_var"):
        assert H.rel_err(sd[k], ost[k]) < 5e-3, k
    assert int(sd["layer1_p.0.bn1.num_batches_tracked"].item()) == 16
    for p in model.parameters():
        assert p.grad is None or torch.isfinite(p.grad).all()


@pytest.mark.parametrize("name,S,N,flat", [("MedT", 128, 4, True), ("gatedaxialunet", 128, 8, False)], ids=["MedT-bs4", "gatedaxialunet-bs8"])
def test_factory_state_train_parity(name, S, N, flat, device):
    """Parity at bench.py's initial state: factory initialisation under torch.manual_seed(3000) BEFORE construction (the reference
    seeds at train.py:118-121, after building the model: its own initial weights are unseeded -- this is a sample of that state), train mode (batch statistics), bench.py's synthetic batch, gates frozen -- against the
    reference's float64 results for exactly that state (tests/golden/factory_*.npz, make_golden.py::factory_fixture).

    MedT 128 bs 4 (BASELINE configs[2], the headline): logits within north_star's FLAT 1e-3 -- no noise scaling.  The
    reference's own float32 runs sit at 2.3e-4 ... 6.1e-4 from its float64 on this state (fixture `logits_noise_runs`).
    gatedaxialunet 128 bs 8 (configs[1]): the reference's own float32 runs are 2.6e-3 ... 4.2e-3 away from its float64 -- a
    flat 1e-3 is not something any float32 evaluation of that network meets (finding, recorded in DESIGN.md section 4); the
    product is held to 1.5 x the reference's float32 deviation there and the figure is printed."""
    fx = H.load_golden(f"factory_{name}_S{S}_N{N}.npz")
    seed = int(fx["meta"][2])
    assert seed == 3000
    torch.manual_seed(seed)
    model = build(name, S, device)                    # same seed, same constructor order -> the reference's initial state
    sd = model.state_dict()
    cs = [sum(v.double().sum().item() for v in sd.values() if v.is_floating_point()),
          sum((v.double() ** 2).sum().item() for v in sd.values() if v.is_floating_point())]
    assert abs(cs[0] - fx["state_checksum"][0]) < 1e-6 * abs(fx["state_checksum"][0]) and \
        abs(cs[1] - fx["state_checksum"][1]) < 1e-6 * fx["state_checksum"][1], "initial state differs from the reference's"
    x, y = H.seeded_input(seed, N, 3, S)
    assert abs(x.double().sum().item() - fx["x_checksum"][0]) < 1e-6
    model.train()
    out = model(x.to(device))
    want = torch.from_numpy(fx["logits"])
    err = H.rel_err(out, want)
    noise = fx["logits_noise_runs"]
    print(f"factory state {name} bs {N} train: logits rel err {err:.2e}; the reference's own float32 runs: "
          f"{noise.min():.2e} ... {noise.max():.2e}")
    assert err < (1e-3 if flat else max(1e-3, 1.5 * float(noise.max()))), err
    loss = torch.nn.functional.cross_entropy(out, y.to(device))
    assert abs(loss.item() - fx["loss"][0]) < 1e-4, (loss.item(), fx["loss"][0])
    loss.backward()
    torch.cuda.synchronize()
    params = dict(model.named_parameters())
    assert {k for k, p in params.items() if p.grad is not None} == set(fx["grad_names"].tolist())      # gates frozen, Q5 tensors unused
    bad, ratios, _ = check_gradient_summaries(fx, params, seed, "train", KNOISE, TOL)
    r = np.sort(np.array([v for v, _ in ratios]))
    print(f"  product error / reference fp32 noise per gradient tensor: median {np.median(r):.2f}, max {r[-1]:.2f} (bound {KNOISE})")
    assert not bad, bad[:8]
