"""The one-launch AxialBlock_wopos forward (medt_wopos_block_fwd, csrc/block_small.hip) against the per-stage path
(medt_conv_block_fwd -> medt_axial_layer_fwd x 2 -> medt_conv_block_fwd) and against a float64 torch restatement of
reference lib/models/axialnet.py:368-391 on the shape it is built for (layer3_p.1-3 of MedT at 128 px, 4 images per group)."""
import copy

import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu


def make_block(device, seed=5):
    import lib as droplib
    torch.manual_seed(seed)
    blk = droplib.models.axialnet.AxialBlock_wopos(128, 64, groups=8, base_width=64, kernel_size=4)
    with torch.no_grad():
        for m in blk.modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.2)
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
    return blk.to(device)


def run(blk, x, dout, fused, training):
    from medt_amd import block, net
    blk = copy.deepcopy(blk)
    blk.train(training)
    old = block.ENABLED
    block.ENABLED = fused
    try:
        xd = x.clone().requires_grad_(True)
        y = net.axial_block_forward(blk, xd, 16)
        (y * dout).sum().backward()
        torch.cuda.synchronize()
    finally:
        block.ENABLED = old
    grads = {k: p.grad.clone() for k, p in blk.named_parameters() if p.grad is not None}
    bufs = {k: v.clone() for k, v in blk.state_dict().items() if "running" in k or "num_batches" in k}
    return y.detach(), xd.grad.clone(), grads, bufs


@pytest.mark.parametrize("training", [True, False], ids=["train", "eval"])
def test_fused_block_equals_stagewise(training, device):
    from medt_amd import _lib, block
    import ctypes
    d = _lib.BlockDesc(64, 128, 64, 4, 4, 8, int(training), 16, 1e-5, 0.1)
    if not block.ENABLED or _lib.lib().medt_wopos_block_workspace_bytes(ctypes.byref(d)) == 0:
        pytest.skip("fused block forward disabled (MEDT_BLOCK_FUSED=0 / MEDT_DISABLE_SMALL=1)")
    blk = make_block(device)
    torch.manual_seed(11)
    x = torch.randn(64, 128, 4, 4, device=device).relu_()          # a block input is a ReLU output
    dout = torch.randn(64, 128, 4, 4, device=device)
    y0, dx0, g0, b0 = run(blk, x, dout, False, training)
    y1, dx1, g1, b1 = run(blk, x, dout, True, training)
    # same arithmetic up to summation order (statistics: wave reductions instead of LDS passes; convolutions: one
    # ascending channel loop instead of four slices)
    assert H.rel_err(y1, y0) < 2e-5, H.rel_err(y1, y0)
    assert H.rel_err(dx1, dx0) < 2e-4, H.rel_err(dx1, dx0)
    assert g0.keys() == g1.keys() and len(g0) >= 18
    gmax = max(v.abs().max().item() for v in g0.values())
    for k in g0:            # (bn_similarity.bias has gradient zero analytically -- the softmax ignores a shift: absolute floor)
        err = (g1[k] - g0[k]).abs().max().item()
        assert err < 5e-4 * max(g0[k].abs().max().item(), 1e-3 * gmax), (k, err, g0[k].abs().max().item())
    for k in b0:
        if "num_batches" in k:
            assert int(b0[k]) == int(b1[k]) == (16 if training else 0), k
        else:
            assert H.rel_err(b1[k], b0[k]) < 1e-5, (k, H.rel_err(b1[k], b0[k]))


@pytest.mark.parametrize("fused", [True, False], ids=["one-launch", "per-stage"])
def test_block_vs_reference_fixture(fused, device):
    """Against the reference itself (tests/golden/block_wopos_*.npz: lib/models/axialnet.py's AxialBlock_wopos applied to two
    patch groups one after the other, float64): outputs in eval and train mode, dx, every parameter gradient, the running
    statistics after the two ordered updates.  1e-3 relative (north_star); observed ~1e-5."""
    import json
    import numpy as np
    import lib as droplib
    from medt_amd import block, net
    from oracle import medt_oracle as O                 # randomize_state only: the fixture's weights are re-derived from the seed
    fx = H.load_golden("block_wopos_C128_P64_S4_G2.npz")
    inplanes, planes, S, groups_n, npg, seed = [int(v) for v in fx["meta"]]
    blk = droplib.models.axialnet.AxialBlock_wopos(inplanes, planes, groups=8, base_width=64, kernel_size=S)
    layout = json.loads(str(fx["state_layout"]))
    assert [k for k, _, _ in layout] == list(blk.state_dict().keys())
    blk.load_state_dict(O.randomize_state({k: v.clone() for k, v in blk.state_dict().items()}, seed))
    blk = blk.to(device)
    x = torch.from_numpy(fx["x"]).float().to(device)
    old = block.ENABLED
    block.ENABLED = fused and old
    if fused and not old:
        pytest.skip("fused block forward disabled")
    try:
        blk.eval()
        with torch.no_grad():
            assert H.rel_err(net.axial_block_forward(blk, x, groups_n), fx["out_eval"]) < 1e-3
        blk.train()
        xg = x.clone().requires_grad_(True)
        y = net.axial_block_forward(blk, xg, groups_n)
        assert H.rel_err(y, fx["out_train"]) < 1e-3
        (y * torch.from_numpy(fx["dout"]).float().to(device)).sum().backward()
        torch.cuda.synchronize()
    finally:
        block.ENABLED = old
    assert H.rel_err(xg.grad, fx["dx"]) < 1e-3
    gscale = max(np.abs(fx[k]).max() for k in fx if k.startswith("grad/"))
    params = dict(blk.named_parameters())
    for k in fx:
        if k.startswith("grad/"):
            want = torch.from_numpy(fx[k])
            scale = max(want.abs().max().item(), 1e-3 * gscale)
            err = (params[k[5:]].grad.double().cpu() - want).abs().max().item() / scale
            assert err < 1e-3, (k, err)
        if k.startswith("buf/"):
            assert H.rel_err(blk.state_dict()[k[4:]].double(), fx[k]) < 1e-3, k


# ---- round 6: the stride-2 FIRST block of a layer with its downsample path as one forward launch (medt_wopos_block_s2_fwd; layer4_p.0
# of MedT at 128 px: 128 -> 128 -> 256 channels, 4x4 -> 2x2 maps).  The per-stage backward runs behind it (adopt mode).
def _s2_block(device):
    import json
    import lib as droplib
    from oracle import medt_oracle as O
    fx = H.load_golden("block_wopos_s2_C128_P128_S4_G2.npz")
    inplanes, planes, S, groups_n, npg, seed, stride = [int(v) for v in fx["meta"]]
    ds = torch.nn.Sequential(droplib.models.axialnet.conv1x1(inplanes, planes * 2, stride), torch.nn.BatchNorm2d(planes * 2))
    blk = droplib.models.axialnet.AxialBlock_wopos(inplanes, planes, stride=stride, downsample=ds, groups=8, base_width=64, kernel_size=S)
    layout = json.loads(str(fx["state_layout"]))
    assert [k for k, _, _ in layout] == list(blk.state_dict().keys())
    blk.load_state_dict(O.randomize_state({k: v.clone() for k, v in blk.state_dict().items()}, seed))
    return fx, blk.to(device), groups_n


@pytest.mark.parametrize("fused", [True, False], ids=["one-launch", "per-stage"])
def test_stride2_block_vs_reference_fixture(fused, device):
    """tests/golden/block_wopos_s2_*.npz: the REFERENCE's AxialBlock_wopos(stride=2, downsample=conv1x1 s2 + BatchNorm) applied to two
    patch groups one after the other (float64): eval and train outputs, dx, all 23 parameter gradients, running statistics."""
    import ctypes
    import numpy as np
    from medt_amd import _lib, block, net
    fx, blk, groups_n = _s2_block(device)
    x = torch.from_numpy(fx["x"]).float().to(device)
    d = _lib.BlockDesc(x.shape[0], 128, 128, 4, 4, 8, 1, groups_n, 1e-5, 0.1)
    if fused and (not block.ENABLED or _lib.lib().medt_wopos_block_s2_workspace_bytes(ctypes.byref(d)) == 0):
        pytest.skip("stride-2 block forward disabled (MEDT_BLOCK_S2=0 / MEDT_BLOCK_FUSED=0)")
    old = block.ENABLED
    block.ENABLED = fused and old
    try:
        blk.eval()
        with torch.no_grad():
            assert H.rel_err(net.axial_block_forward(blk, x, groups_n), fx["out_eval"]) < 1e-3
        blk.train()
        xg = x.clone().requires_grad_(True)
        y = net.axial_block_forward(blk, xg, groups_n)
        e = H.rel_err(y, fx["out_train"])
        assert e < 1e-3, e
        (y * torch.from_numpy(fx["dout"]).float().to(device)).sum().backward()
        torch.cuda.synchronize()
    finally:
        block.ENABLED = old
    assert H.rel_err(xg.grad, fx["dx"]) < 1e-3
    gscale = max(np.abs(fx[k]).max() for k in fx if k.startswith("grad/"))
    params = dict(blk.named_parameters())
    n = 0
    for k in fx:
        if k.startswith("grad/"):
            want = torch.from_numpy(fx[k])
            scale = max(want.abs().max().item(), 1e-3 * gscale)
            err = (params[k[5:]].grad.double().cpu() - want).abs().max().item() / scale
            assert err < 1e-3, (k, err)
            n += 1
        if k.startswith("buf/"):
            assert H.rel_err(blk.state_dict()[k[4:]].double(), fx[k]) < 1e-3, k
    assert n == 23


def test_stride2_block_is_one_forward_launch(device):
    from medt_amd import block, net
    from medt_amd.defer import StepQueue
    fx, blk, groups_n = _s2_block(device)
    if not block.ENABLED or block.fused_forward(blk.train(), torch.from_numpy(fx["x"]).float().to(device), groups_n) is None:
        pytest.skip("stride-2 block forward disabled")
    q = StepQueue()
    with q.active():
        with torch.no_grad():
            net.axial_block_forward(blk, torch.from_numpy(fx["x"]).float().to(device), groups_n)
        assert q.pending() == 9                 # nine BatchNorm bookkeeping jobs from ONE call; the adopting stages record nothing
    torch.cuda.synchronize()


def test_fused_block_is_taken_and_counts_one_launch(device):
    """The block really runs as one forward launch: eight BatchNorm bookkeeping jobs are recorded by a single call, and the
    adopt-mode stages launch nothing (their workspaces are never requested)."""
    from medt_amd import block, net
    from medt_amd.defer import StepQueue
    if not block.ENABLED:
        pytest.skip("fused block forward disabled")
    blk = make_block(device).train()
    x = torch.randn(64, 128, 4, 4, device=device).relu_()
    q = StepQueue()
    with q.active():
        with torch.no_grad():
            net.axial_block_forward(blk, x, 16)
        assert q.pending() == 8
    torch.cuda.synchronize()


# ---- the one-launch BACKWARD (medt_wopos_block_bwd): the default since round 5 (first MI355X run green; MEDT_BLOCK_BWD=0 for
# the whole process -- the library reads the variable once -- switches it off).  test_block_vs_reference_fixture[one-launch]
# above runs through it as well.
bwd_opt_in = pytest.mark.skipif(__import__("os").environ.get("MEDT_BLOCK_BWD", "1") == "0",
                                reason="one-launch block backward switched off: MEDT_BLOCK_BWD=0")


@bwd_opt_in
@pytest.mark.parametrize("training", [True, False], ids=["train", "eval"])
def test_block_backward_one_launch_equals_stagewise(training, device):
    from medt_amd import block
    assert block.BWD_ENABLED
    blk = make_block(device)
    torch.manual_seed(11)
    x = torch.randn(64, 128, 4, 4, device=device).relu_()
    dout = torch.randn(64, 128, 4, 4, device=device)
    block.BWD_ENABLED = False
    try:
        y0, dx0, g0, b0 = run(blk, x, dout, True, training)          # one-launch forward, per-stage backward
    finally:
        block.BWD_ENABLED = True
    y1, dx1, g1, b1 = run(blk, x, dout, True, training)              # one autograd node, one-launch backward
    assert torch.equal(y1, y0)                                        # the same forward launch
    assert H.rel_err(dx1, dx0) < 2e-5, H.rel_err(dx1, dx0)
    assert g0.keys() == g1.keys() and len(g0) == 20
    gmax = max(v.abs().max().item() for v in g0.values())
    for k in g0:
        err = (g1[k] - g0[k]).abs().max().item()
        assert err < 5e-5 * max(g0[k].abs().max().item(), 1e-2 * gmax), (k, err, g0[k].abs().max().item())
    for k in b0:
        assert torch.equal(b0[k], b1[k]), k


@bwd_opt_in
def test_block_backward_is_one_launch_and_takes_the_deposit(device):
    """One call records the block's 4 + 4 parameter-gradient jobs (2 BatchNorm finalisations, 2 layer finalisations, 4 weight
    gradients), and what another consumer of x deposited in x's sink comes back inside dx."""
    from medt_amd import block, net, ops
    from medt_amd.defer import StepQueue
    blk = make_block(device).train()
    torch.manual_seed(3)
    x = torch.randn(64, 128, 4, 4, device=device).relu_().requires_grad_(True)
    dout = torch.randn(64, 128, 4, 4, device=device)
    y = net.axial_block_forward(blk, x, 16)
    assert type(y.grad_fn).__name__ == "WoposBlockFnBackward"
    (y * dout).sum().backward()
    dx_plain = x.grad.clone()
    x.grad = None
    for p in blk.parameters():
        p.grad = None
    del x._medt_sink                              # (the first backward closed x's sink: role "final")
    y = net.axial_block_forward(blk, x, 16)
    extra = torch.randn_like(x)
    assert ops.sink_of(x).deposit(extra)
    (y * dout).sum().backward()
    torch.cuda.synchronize()
    assert H.rel_err(x.grad, dx_plain + extra) < 1e-6


# ---- the 8x8-map block kernels (layer2_p.1 of MedT-128): the default since round 5 (MEDT_BLOCK8=0 switches them off)
block8_opt_in = pytest.mark.skipif(__import__("os").environ.get("MEDT_BLOCK8", "1") == "0",
                                   reason="8x8-map one-launch block kernels switched off: MEDT_BLOCK8=0")


@block8_opt_in
@pytest.mark.parametrize("training", [True, False], ids=["train", "eval"])
def test_block8_fused_equals_stagewise(training, device):
    """wopos_block8_fwd_kernel (a wave = one image x a quarter of the channels; BatchNorm statistics merged across the four
    image-waves) against the four per-stage launches on layer2_p.1's shape; verified on the CPU lane emulator against the oracle
    (tests/test_lane_emu.py::test_block8_forward_kernel_on_the_emulator)."""
    import lib as droplib
    from medt_amd import _lib, block
    import ctypes
    d = _lib.BlockDesc(64, 64, 32, 8, 8, 8, int(training), 16, 1e-5, 0.1)
    assert _lib.lib().medt_wopos_block_workspace_bytes(ctypes.byref(d)) > 0
    torch.manual_seed(6)
    blk = droplib.models.axialnet.AxialBlock_wopos(64, 32, groups=8, base_width=64, kernel_size=8)
    with torch.no_grad():
        for m in blk.modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.2)
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
    blk = blk.to(device)
    x = torch.randn(64, 64, 8, 8, device=device).relu_()
    dout = torch.randn(64, 64, 8, 8, device=device)
    y0, dx0, g0, b0 = run(blk, x, dout, False, training)
    y1, dx1, g1, b1 = run(blk, x, dout, True, training)
    assert H.rel_err(y1, y0) < 2e-5, H.rel_err(y1, y0)
    assert H.rel_err(dx1, dx0) < 2e-4, H.rel_err(dx1, dx0)
    assert g0.keys() == g1.keys() and len(g0) == 20
    gmax = max(v.abs().max().item() for v in g0.values())
    for k in g0:
        err = (g1[k] - g0[k]).abs().max().item()
        assert err < 5e-4 * max(g0[k].abs().max().item(), 1e-3 * gmax), (k, err)
    for k in b0:
        if "num_batches" in k:
            assert int(b0[k]) == int(b1[k])
        else:
            assert H.rel_err(b1[k], b0[k]) < 1e-5, k


@block8_opt_in
@bwd_opt_in
@pytest.mark.parametrize("training", [True, False], ids=["train", "eval"])
def test_block8_backward_one_launch_equals_stagewise(training, device):
    """wopos_block8_bwd_kernel (default since round 5) against the six per-stage backward launches on layer2_p.1's shape;
    verified on the CPU lane emulator against the oracle (tests/test_lane_emu.py::test_block8_backward_kernel_on_the_emulator)."""
    import lib as droplib
    from medt_amd import block
    torch.manual_seed(6)
    blk = droplib.models.axialnet.AxialBlock_wopos(64, 32, groups=8, base_width=64, kernel_size=8)
    with torch.no_grad():
        for m in blk.modules():
            if isinstance(m, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.2)
                m.running_mean.normal_(0, 0.2)
                m.running_var.uniform_(0.5, 1.5)
    blk = blk.to(device)
    x = torch.randn(64, 64, 8, 8, device=device).relu_()
    dout = torch.randn(64, 64, 8, 8, device=device)
    block.BWD_ENABLED = False
    try:
        y0, dx0, g0, b0 = run(blk, x, dout, True, training)
    finally:
        block.BWD_ENABLED = True
    y1, dx1, g1, b1 = run(blk, x, dout, True, training)
    assert torch.equal(y1, y0)
    assert H.rel_err(dx1, dx0) < 2e-5, H.rel_err(dx1, dx0)
    assert g0.keys() == g1.keys() and len(g0) == 20
    gmax = max(v.abs().max().item() for v in g0.values())
    for k in g0:
        err = (g1[k] - g0[k]).abs().max().item()
        assert err < 5e-5 * max(g0[k].abs().max().item(), 1e-2 * gmax), (k, err)
