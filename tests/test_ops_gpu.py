"""HIP ops around the attention layers (C ABI) vs fp64 torch on the CPU.  Tolerance 2e-4 relative
(these are well-conditioned single ops; observed errors are ~1e-6)."""
import copy

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

import helpers as H

pytestmark = pytest.mark.gpu
TOL = 2e-4


def _cmp(a, b, tol=TOL, what=""):
    err = H.rel_err(a, b)
    assert err < tol, (what, err)


def ref_conv_block(x, conv, bn, res, relu, training, groups):
    """fp64 CPU reference; bn_groups realised as chunked batch_norm with sequential running-stat updates."""
    z = F.conv2d(x, conv.weight, conv.bias, stride=conv.stride, padding=conv.padding)
    if bn is not None:
        outs = []
        for zc in z.chunk(groups if training else 1, 0):
            outs.append(F.batch_norm(zc, bn.running_mean, bn.running_var, bn.weight, bn.bias, training, bn.momentum, bn.eps))
            if training:
                bn.num_batches_tracked += 1
        z = torch.cat(outs, 0)
    if res is not None:
        z = z + res
    return F.relu(z) if relu else z


CONV_CASES = [
    # Cin, Cout, K, stride, pad, bias, bn, res, relu, N, H, groups
    (3, 8, 7, 2, 3, False, True, False, True, 2, 32, 1),        # stem conv1
    (3, 64, 7, 2, 3, False, True, False, True, 4, 32, 4),       # conv1_p, grouped BN (round 5: the LDS-patch MFMA stem kernel)
    (3, 40, 7, 2, 3, True, False, False, True, 3, 64, 1),       # stem kernel on 32-wide output maps (two tiles per row), partial channel tile, bias
    (3, 96, 7, 2, 3, False, True, False, False, 2, 32, 2),      # stem kernel, two channel tiles (64 + 32), no ReLU
    (8, 128, 3, 1, 1, False, True, False, True, 2, 16, 1),      # conv2
    (128, 8, 3, 1, 1, False, True, False, True, 2, 16, 2),      # conv3
    (32, 16, 1, 1, 0, False, True, False, True, 2, 16, 1),      # conv_down
    (16, 32, 1, 1, 0, False, True, True, True, 4, 8, 2),        # conv_up + identity + relu
    (32, 64, 1, 2, 0, False, True, False, False, 2, 16, 1),     # downsample (stride 2, no relu)
    (64, 64, 3, 2, 1, True, False, False, False, 2, 4, 1),      # decoder1 (stride 2, bias)
    (32, 16, 3, 1, 1, True, False, False, False, 2, 16, 1),     # decoder
    (16, 16, 3, 1, 1, True, False, False, True, 2, 16, 1),      # decoderf + relu
    (16, 2, 1, 1, 0, True, False, False, False, 2, 16, 1),      # adjust
    (256, 256, 3, 2, 1, True, False, False, False, 8, 2, 1),    # decoder1_p on 2x2 maps
    (20, 24, 3, 1, 1, False, True, False, True, 3, 9, 1),       # odd sizes
    # shapes that take the fp32-MFMA implicit-GEMM path (3x3, Cin*9 >= 256, >= 128 output tiles)
    (64, 128, 3, 1, 1, False, True, False, True, 32, 16, 4),    # conv2_p with grouped BN
    (128, 64, 3, 1, 1, True, False, False, True, 32, 16, 1),    # conv3_p-shaped, bias + relu, no BN
    (32, 64, 3, 2, 1, True, False, False, False, 16, 64, 1),    # stride 2 (forward + weight gradient on MFMA)
    (40, 72, 3, 1, 1, False, True, True, True, 36, 15, 2),      # ragged tiles: Cout % 64 != 0, positions % 64 != 0
    (48, 80, 3, 1, 1, False, True, True, True, 32, 16, 2),      # 16-wide-map kernels with a partial channel tile (fwd / dgrad / wgrad)
    # deep contraction, narrow side: the wave-split kernels (four waves share the channel loop, 64-position workgroups)
    (128, 8, 3, 1, 1, False, True, False, True, 4, 64, 1),      # conv3 at BASELINE size (forward wave-split; dgrad wide)
    (8, 128, 3, 1, 1, False, True, False, True, 4, 64, 1),      # conv2 at BASELINE size (dgrad wave-split at 16384 positions)
    (64, 32, 3, 1, 1, True, False, False, False, 4, 32, 1),     # decoder4: bias, no BatchNorm
    (72, 24, 3, 2, 1, False, True, True, True, 6, 18, 3),       # ragged: stride 2, 81-position maps, grouped statistics
    # round 6: the thin-channel LDS-patch MFMA kernel (conv3x3_thin_fwd_kernel: forward and, on the flipped weights, backward-data) --
    # the four BASELINE-size cases above (conv2 / conv3 at 64 x 64, decoder4) run it too
    (32, 16, 3, 1, 1, True, False, False, False, 4, 64, 1),     # decoder5 at BASELINE size (one row per workgroup, four column tiles)
    (16, 16, 3, 1, 1, True, False, False, True, 4, 128, 1),     # decoderf at BASELINE size (four rows per wave, two column workgroups per row band)
    (32, 16, 3, 1, 1, True, False, False, False, 64, 16, 1),    # decoder5_p: 16-wide maps (one column tile, four row groups)
    (16, 24, 3, 1, 1, False, True, True, True, 4, 32, 2),       # 32-wide maps (two column tiles x two row groups), grouped BatchNorm, residual, Cout % 16 != 0
    (8, 40, 3, 1, 1, False, True, False, True, 2, 64, 1),       # 8-channel chunks, three row blocks over two workgroups (the last one half empty)
    (8, 128, 3, 1, 1, False, True, False, True, 2, 64, 1),      # conv2 at 2 images: two rows per wave (the plan follows the workgroup count)
    (16, 16, 3, 1, 1, True, False, False, True, 2, 128, 1),     # decoderf at 2 images: two rows per wave, 16-channel chunk
    (32, 48, 3, 1, 1, False, True, False, False, 1, 64, 1),     # two row blocks x two rows per wave
    (128, 8, 3, 1, 1, False, True, False, True, 2, 64, 1),      # conv3 / decoder5 / decoder5_p at the 2-image fixtures' sizes
    (32, 16, 3, 1, 1, True, False, False, False, 2, 64, 1),
    (32, 16, 3, 1, 1, True, False, False, False, 32, 16, 1),
    # MedT's local branch at BASELINE size (16 patch groups x 4 images): the BatchNorm backward + 1x1 dgrad of these blocks is
    # ONE launch (bn_dgrad1x1_small_kernel: 256 / 512 threads, 1 / 2 / 4 input channels per thread)
    (128, 64, 1, 1, 0, False, True, False, True, 64, 4, 16),    # layer3_p.1-3 conv_down (T=256, 16 values per thread)
    (64, 128, 1, 1, 0, False, True, True, True, 64, 4, 16),     # layer3_p.1-3 conv_up + identity
    (64, 32, 1, 1, 0, False, True, False, True, 64, 8, 16),     # layer2_p.1 conv_down (one thread group over the positions)
    (32, 64, 1, 1, 0, False, True, True, True, 64, 8, 16),      # layer2_p.1 conv_up (T=512)
    (64, 64, 1, 1, 0, False, True, False, True, 64, 8, 16),     # layer3_p.0 conv_down (T=512, 4 channels per thread)
    (128, 256, 1, 1, 0, False, True, True, True, 64, 2, 16),    # layer4_p.0 conv_up on 2x2 maps (one thread per channel)
    (128, 128, 1, 1, 0, False, True, False, True, 64, 4, 16),   # layer4_p.0 conv_down
    # more local-branch shapes: 1024-position groups, 8 input channels, stride-2 downsample
    (8, 16, 1, 1, 0, False, True, False, True, 64, 16, 16),     # layer1_p.0 conv_down (1024 positions, one channel per wave)
    (16, 32, 1, 1, 0, False, True, True, True, 64, 16, 16),     # layer1_p.0 conv_up + downsampled identity
    (32, 32, 1, 1, 0, False, True, False, True, 64, 16, 16),    # layer2_p.0 conv_down
    (32, 64, 1, 2, 0, False, True, False, False, 64, 16, 16),   # layer2_p.0 downsample (stride 2, 256 output positions)
    (64, 128, 1, 2, 0, False, True, False, False, 64, 8, 16),   # layer3_p.0 downsample (stride 2, 64 output positions)
    # round 6: 32768-value BatchNorm populations (gatedaxialunet bs 8 at 64 x 64, MedT-256 bs 2 at 128 x 128) on the one-launch backward
    # (bn_act_bwd_chan_kernel<1024, 8>: eight float4s per thread) instead of statistics -> finalisation -> application
    (8, 16, 1, 1, 0, False, True, False, True, 8, 64, 1),       # layer1.0 conv_down of gatedaxialunet at bs 8
    (16, 8, 1, 1, 0, False, True, True, True, 2, 128, 1),       # the same population from 2 images of 128 x 128, with the identity
    # round 6: deep thin contractions at 8 images of 64 x 64 / 2 of 128 x 128: one row per wave + four K-groups (conv_thin_plan's rule for >= 4 chunks)
    (128, 8, 3, 1, 1, False, True, False, True, 8, 64, 1),      # conv3 of gatedaxialunet at bs 8 (forward <16,1,1,4> on 512 workgroups)
    (8, 128, 3, 1, 1, False, True, False, True, 8, 64, 1),      # conv2 at bs 8 (its backward-data is the deep one)
    (128, 8, 3, 1, 1, False, True, False, True, 2, 128, 1),     # conv3 of MedT-256 at bs 2
    (64, 32, 3, 1, 1, True, False, False, False, 8, 32, 1),     # decoder4 at bs 8: 128 MFMA tiles -> split in two k-slices (round 6)
    (64, 48, 3, 1, 1, False, True, False, True, 8, 32, 1),      # the same tile count with BatchNorm (statistics from the split-K epilogue)
    (256, 256, 3, 2, 1, True, False, False, False, 64, 2, 1),   # decoder1_p at BASELINE size: stride 2 on 2 x 2 maps (staged wave-split backward-data, 4 of 9 taps live)
    (256, 256, 3, 2, 1, True, False, False, False, 8, 4, 1),    # decoder1 of the unets at bs 8: 4 x 4 -> 2 x 2 (every tap live somewhere)
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "-".join(str(int(v)) for v in c))
@pytest.mark.parametrize("training", [True, False], ids=["train", "eval"])
def test_conv_block(case, training, device):
    from medt_amd import ops
    Cin, Cout, K, stride, pad, bias, has_bn, has_res, relu, N, S, groups = case
    torch.manual_seed(Cin * 100 + Cout + K)
    conv = nn.Conv2d(Cin, Cout, K, stride=stride, padding=pad, bias=bias)
    bn = nn.BatchNorm2d(Cout) if has_bn else None
    if bn is not None:
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_(0, 0.2)
            bn.running_mean.normal_(0, 0.2)
            bn.running_var.uniform_(0.5, 1.5)
    x = torch.randn(N, Cin, S, S)
    So = (S + 2 * pad - K) // stride + 1
    res = torch.randn(N, Cout, So, So) if has_res else None
    dout = torch.randn(N, Cout, So, So)
    # reference (fp64, CPU)
    conv64 = copy.deepcopy(conv).double()
    bn64 = copy.deepcopy(bn).double() if bn is not None else None
    x64 = x.double().requires_grad_(True)
    r64 = res.double().requires_grad_(True) if has_res else None
    y64 = ref_conv_block(x64, conv64, bn64, r64, relu, training, groups)
    (y64 * dout.double()).sum().backward()
    # product
    convd = copy.deepcopy(conv).to(device)
    bnd = copy.deepcopy(bn).to(device) if bn is not None else None
    xd = x.to(device).requires_grad_(True)
    rd = res.to(device).requires_grad_(True) if has_res else None
    y = ops.conv_block(xd, convd, bnd, rd, relu, training, groups if training else 1)
    (y * dout.to(device)).sum().backward()
    torch.cuda.synchronize()
    _cmp(y, y64, what="y")
    _cmp(xd.grad, x64.grad, what="dx")
    _cmp(convd.weight.grad, conv64.weight.grad, what="dw")
    if bias:
        _cmp(convd.bias.grad, conv64.bias.grad, what="dbias")
    if has_res:
        _cmp(rd.grad, r64.grad, what="dres")
    if bn is not None:
        gs = max(bn64.weight.grad.abs().max().item(), bn64.bias.grad.abs().max().item())
        assert (bnd.weight.grad.double().cpu() - bn64.weight.grad).abs().max().item() < TOL * gs
        assert (bnd.bias.grad.double().cpu() - bn64.bias.grad).abs().max().item() < TOL * gs
        if training:
            _cmp(bnd.running_mean, bn64.running_mean, what="running_mean")
            _cmp(bnd.running_var, bn64.running_var, what="running_var")
            assert int(bnd.num_batches_tracked.item()) == int(bn64.num_batches_tracked.item())


@pytest.mark.parametrize("shape", [(2, 3, 1, 1), (2, 4, 2, 2), (1, 5, 7, 3), (2, 16, 16, 16), (1, 2, 64, 64)])
@pytest.mark.parametrize("with_skip", [True, False])
def test_up2x_relu_add(shape, with_skip, device):
    from medt_amd import ops
    torch.manual_seed(shape[2])
    x = torch.randn(shape)
    N, C, Hh, Ww = shape
    skip = torch.randn(N, C, 2 * Hh, 2 * Ww) if with_skip else None
    dout = torch.randn(N, C, 2 * Hh, 2 * Ww)
    x64 = x.double().requires_grad_(True)
    s64 = skip.double().requires_grad_(True) if with_skip else None
    y64 = F.relu(F.interpolate(x64, scale_factor=(2, 2), mode="bilinear"))
    if with_skip:
        y64 = y64 + s64
    (y64 * dout.double()).sum().backward()
    xd = x.to(device).requires_grad_(True)
    sd = skip.to(device).requires_grad_(True) if with_skip else None
    y = ops.up2x_relu_add(xd, sd)
    (y * dout.to(device)).sum().backward()
    _cmp(y, y64, 1e-5, "y")
    _cmp(xd.grad, x64.grad, 1e-5, "dx")
    if with_skip:
        _cmp(sd.grad, s64.grad, 1e-6, "dskip")


@pytest.mark.parametrize("S", [128, 256])
def test_patch_gather_and_merge(S, device):
    from medt_amd import ops
    torch.manual_seed(S)
    N, C = 2, 5
    img = torch.randn(N, 3, S, S)
    xp = ops.patch_gather(img.to(device))
    want = torch.cat([img[:, :, 32 * i:32 * i + 32, 32 * j:32 * j + 32] for i in range(4) for j in range(4)], 0)
    assert torch.equal(xp.cpu(), want)
    x = torch.randn(N, C, S, S)
    yp = torch.randn(16 * N, C, 32, 32)
    dout = torch.randn(N, C, S, S)
    x64 = x.double().requires_grad_(True)
    p64 = yp.double().requires_grad_(True)
    loc = x64.clone()
    for p in range(16):
        i, j = divmod(p, 4)
        loc[:, :, 32 * i:32 * i + 32, 32 * j:32 * j + 32] = p64[p * N:(p + 1) * N]
    y64 = x64 + loc
    (y64 * dout.double()).sum().backward()
    xd = x.to(device).requires_grad_(True)
    pd = yp.to(device).requires_grad_(True)
    y = ops.logo_merge(xd, pd)
    (y * dout.to(device)).sum().backward()
    _cmp(y, y64, 1e-6)
    _cmp(xd.grad, x64.grad, 1e-6)
    _cmp(pd.grad, p64.grad, 1e-6)


def test_cross_entropy(device):
    import medt_amd
    torch.manual_seed(0)
    logits = torch.randn(3, 2, 17, 19) * 3
    target = torch.randint(0, 2, (3, 17, 19))
    target[0, 0, :5] = -100
    l64 = logits.double().requires_grad_(True)
    loss64 = F.cross_entropy(l64, target)
    (loss64 * 1.7).backward()
    ld = logits.to(device).requires_grad_(True)
    loss = medt_amd.cross_entropy(ld, target.to(device))
    (loss * 1.7).backward()
    assert abs(loss.item() - loss64.item()) < 1e-5
    _cmp(ld.grad, l64.grad, 1e-5)


def test_flat_adam_matches_torch_adam(device):
    from medt_amd.optim import FlatAdam
    torch.manual_seed(0)
    shapes = [(7, 3), (), (5,), (2, 3, 4, 4)]
    ps = [torch.randn(s) for s in shapes]
    ref = [nn.Parameter(p.clone().double()) for p in ps]
    mine = [nn.Parameter(p.clone().to(device)) for p in ps]
    unused_ref, unused_mine = nn.Parameter(torch.ones(3).double()), nn.Parameter(torch.ones(3, device=device))
    late_ref, late_mine = nn.Parameter(torch.tensor(0.1).double()), nn.Parameter(torch.tensor(0.1, device=device))
    o_ref = torch.optim.Adam(ref + [unused_ref, late_ref], lr=1e-2, weight_decay=1e-2)
    o_mine = FlatAdam(mine + [unused_mine, late_mine], lr=1e-2, weight_decay=1e-2)
    for step in range(6):
        o_ref.zero_grad()
        o_mine.zero_grad()
        for a, b in zip(ref, mine):
            g = torch.randn(a.shape)
            a.grad = g.double()
            b.grad = g.to(device)
        if step >= 3:                                   # the gates join at "epoch 10" with a fresh step counter
            g = torch.randn(())
            late_ref.grad = g.double()
            late_mine.grad = g.to(device)
        o_ref.step()
        o_mine.step()
    for a, b in zip(ref + [unused_ref, late_ref], mine + [unused_mine, late_mine]):
        _cmp(b, a, 1e-5)


def test_seg_counts_and_scores(device):
    """Device-side confusion counts + the scoring conventions of performancemetrics_monuseg.m (per-pixel loops)."""
    import medt_amd
    import metrics
    torch.manual_seed(4)
    logits = torch.randn(5, 2, 37, 41)
    logits[1, 1] = -3.0                                   # an image without any predicted foreground (tp = 0 -> scores 1)
    logits[2, 1, 0, 0] = 0.5                              # threshold is inclusive (>= 0.5, test.py)
    target = torch.randint(0, 2, (5, 37, 41))
    target[3] = 0
    counts = medt_amd.seg_counts(logits.to(device), target.to(device)).cpu()
    pred, gt = logits[:, 1] >= 0.5, target > 0
    want = torch.stack([(pred & gt).flatten(1).sum(1), (pred & ~gt).flatten(1).sum(1), (~pred & gt).flatten(1).sum(1),
                        (~pred & ~gt).flatten(1).sum(1)], 1).int()
    assert torch.equal(counts, want)
    f1, iou, pa = metrics.segmentation_scores(counts)
    for n in range(5):                                    # the MATLAB loop, literally
        tp = fp = fn = uni = ttp = 0
        for p_, g_ in zip(pred[n].flatten().tolist(), gt[n].flatten().tolist()):
            if not p_:
                if g_:
                    fp += 1; uni += 1; ttp += 1           # (their "fp" is a missed foreground pixel)
            else:
                if g_:
                    tp += 1; ttp += 1
                else:
                    fn += 1
                uni += 1
        if tp:
            assert abs(f1[n].item() - 2 * tp / (2 * tp + fp + fn)) < 1e-12
            assert abs(iou[n].item() - tp / uni) < 1e-12
            assert abs(pa[n].item() - tp / ttp) < 1e-12
        else:
            assert f1[n].item() == iou[n].item() == pa[n].item() == 1.0


def test_cross_entropy_rejects_out_of_range_targets(device):
    """F.cross_entropy raises on class indices outside [0, K) that are not ignore_index (a mask left at 255): the
    kernel counts them and the host raises; ignore_index itself stays legal."""
    import medt_amd
    logits = torch.randn(2, 2, 8, 8, device=device)
    target = torch.randint(0, 2, (2, 8, 8), device=device)
    target[0, 0, 0] = -100
    medt_amd.cross_entropy(logits, target)                      # fine: ignored pixel
    target[1, 3, 3] = 255
    with pytest.raises(medt_amd.MedtError):
        medt_amd.cross_entropy(logits, target)


def test_batchnorm_momentum_none_is_refused(device):
    import medt_amd
    from medt_amd import ops
    conv = nn.Conv2d(3, 4, 1, bias=False).to(device)
    bn = nn.BatchNorm2d(4, momentum=None).to(device)
    with pytest.raises(medt_amd.MedtError):
        ops.conv_block(torch.randn(2, 3, 4, 4, device=device), conv, bn, training=True)


def test_single_value_training_batchnorm_is_refused(device):
    """nn.BatchNorm raises "Expected more than 1 value per channel when training"; so does the conv + BN block."""
    import medt_amd
    from medt_amd import ops
    conv = nn.Conv2d(3, 4, 1, bias=False).to(device)
    bn = nn.BatchNorm2d(4).to(device)
    with pytest.raises(medt_amd.MedtError):
        ops.conv_block(torch.randn(1, 3, 1, 1, device=device), conv, bn, training=True)
    ops.conv_block(torch.randn(1, 3, 1, 1, device=device), conv, bn, training=False)      # eval: fine


@pytest.mark.parametrize("k_final", [1, 3])
def test_gradient_fan_in_through_sink(k_final, device):
    """Two consumers of one tensor share a GradSink (medt_amd.ops): the later-created one deposits its input gradient,
    the earlier-created one adds it -- in the epilogue of its dgrad kernel when it is 1x1, with an explicit add when it
    is wider -- and returns the sum.  Either way the tensor's gradient equals autograd's plain sum."""
    from medt_amd import ops
    torch.manual_seed(5)
    a = nn.Conv2d(8, 12, k_final, padding=k_final // 2, bias=True).to(device)
    b = nn.Conv2d(8, 6, 1, bias=True).to(device)
    x0 = torch.randn(2, 8, 16, 16, device=device)
    ga, gb = torch.randn(2, 12, 16, 16, device=device), torch.randn(2, 6, 16, 16, device=device)

    def run(shared):
        x = (x0 * 1.0).requires_grad_(True)                  # a non-leaf producer output, like a layer's x1..x3
        x.retain_grad()
        s = ops.sink_of(x) if shared else None
        ya = ops.conv_block(x, a, x_sink=s, x_role="final" if shared else None)
        yb = ops.conv_block(x, b, x_sink=s, x_role="deposit" if shared else None)
        ((ya * ga).sum() + (yb * gb).sum()).backward()
        if shared:
            assert s.pending is None                          # nothing stranded
        return x.grad.clone()

    _cmp(run(True), run(False), 1e-6, "fan-in")


# The recorded (grouped) weight gradients the way a training step takes them: parameter gradients in FlatAdam's slots, the
# weight-gradient jobs recorded into a StepQueue and issued by its flush -- conv_wgrad_mfma_grouped_kernel.  Round 5: 1x1 stride-1
# problems with 16-byte-aligned rows take the 16-byte position-axis body (conv_wgrad_v4_body32<K>: 128-position steps; K = 3 with stride 1 / pad 1 on maps whose width is a multiple of 4 likewise, chunks of
# 256 ... 1024 positions); everything else the scalar-load bodies.  Against fp64 torch.
RECORDED_CASES = [
    # Cin, Cout, K, stride, pad, bn, N, S, groups
    (32, 16, 1, 1, 0, True, 2, 16, 1),        # conv_down: one chunk pair, BatchNorm backward applied on load
    (16, 32, 1, 1, 0, True, 4, 8, 2),         # two BatchNorm groups inside one chunk (per-lane coefficient group)
    (20, 24, 1, 1, 0, True, 3, 6, 3),         # ragged: 20 / 24 channels, 108 positions (a partial 128-position step), 3 groups
    (72, 40, 1, 1, 0, False, 2, 10, 1),       # two k-tiles (72 input channels), 40 output channels (a partial row block), no BN
    (128, 64, 1, 1, 0, True, 64, 4, 16),      # layer3_p.1-3 conv_down at BASELINE size: 1024 positions = 4 chunks of 256
    (16, 32, 1, 1, 0, True, 16, 32, 1),       # 16384 positions: 16 chunks of 1024 (8 steps each)
    (8, 8, 1, 1, 0, True, 2, 3, 1),           # 3x3 maps: 9 positions per image, not a multiple of 4 -> the scalar-load body
    (32, 64, 1, 2, 0, True, 2, 16, 1),        # stride 2 -> the scalar-load body
    (16, 16, 3, 1, 1, True, 2, 8, 1),         # 3x3 stride 1 pad 1 on 8-wide maps: the 16-byte body's K = 3 instance (one float4 per row)
    (16, 16, 3, 1, 1, False, 4, 128, 1),      # decoderf at BASELINE size: 65536 positions = 256 chunks / slabs
    (16, 2, 1, 1, 0, False, 4, 128, 1),       # adjust: 65536 positions = 256 chunks
    (32, 16, 3, 1, 1, True, 3, 12, 3),        # 3x3, 12-wide maps (three float4 per row: left / interior / right borders), 3 BatchNorm groups
    (128, 8, 3, 1, 1, True, 2, 16, 1),        # conv3's shape: 1152 = 18 k-tiles of (channel, tap) rows, 8 output channels
    (8, 40, 3, 1, 1, False, 2, 20, 1),        # 72 (channel, tap) rows = a full and a partial k-tile, 40 output channels
    (16, 16, 3, 1, 1, True, 2, 6, 1),         # 6-wide maps: not a multiple of 4 -> the scalar-load body
    (16, 16, 3, 2, 1, True, 2, 8, 1),         # stride 2 -> the scalar-load body
]


@pytest.mark.parametrize("case", RECORDED_CASES, ids=lambda c: "-".join(str(int(v)) for v in c))
def test_recorded_weight_gradient(case, device):
    from medt_amd import ops
    from medt_amd.defer import StepQueue
    from medt_amd.optim import FlatAdam
    Cin, Cout, K, stride, pad, has_bn, N, S, groups = case
    torch.manual_seed(Cin * 100 + Cout + K + S)
    conv = nn.Conv2d(Cin, Cout, K, stride=stride, padding=pad, bias=False)
    bn = nn.BatchNorm2d(Cout) if has_bn else None
    if bn is not None:
        with torch.no_grad():
            bn.weight.uniform_(0.5, 1.5)
            bn.bias.normal_(0, 0.2)
    x = torch.randn(N, Cin, S, S)
    So = (S + 2 * pad - K) // stride + 1
    dout = torch.randn(N, Cout, So, So)
    conv64, bn64 = copy.deepcopy(conv).double(), (copy.deepcopy(bn).double() if bn is not None else None)
    y64 = ref_conv_block(x.double(), conv64, bn64, None, True, True, groups)
    (y64 * dout.double()).sum().backward()
    convd, bnd = copy.deepcopy(conv).to(device), (copy.deepcopy(bn).to(device) if bn is not None else None)
    params = [convd.weight] + ([bnd.weight, bnd.bias] if bnd is not None else [])
    opt = FlatAdam(params, lr=0.0)
    xd, dd = x.to(device), dout.to(device)
    if device.type == "cpu":                              # the emulated device (pytest --emulate)
        from emu_device import DeviceTensor
        xd, dd = xd.as_subclass(DeviceTensor), dd.as_subclass(DeviceTensor)

    def fwd_bwd(q=None):
        opt.zero_grad()
        y = ops.conv_block(xd, convd, bnd, None, True, True, groups)
        if q is not None:
            q.flush()                                     # the forward's recorded statistics: backward reads them (TrainStep does this)
        (y * dd).sum().backward()
    fwd_bwd()                                            # adoption step: gradients through autograd, immediate launches
    opt.pack_gradients()
    immediate = convd.weight.grad.detach().clone()
    q = StepQueue()
    with q.active():
        fwd_bwd(q)
        assert q.pending() > 0                            # the weight gradient (and its slab reduction) were recorded
    opt.pack_gradients()
    if device.type == "cuda":
        torch.cuda.synchronize()
    _cmp(convd.weight.grad, conv64.weight.grad, what="recorded dw")
    _cmp(convd.weight.grad, immediate, tol=2e-5, what="recorded vs immediate dw")
    if bnd is not None:
        gs = max(bn64.weight.grad.abs().max().item(), bn64.bias.grad.abs().max().item())
        assert (bnd.weight.grad.double().cpu() - bn64.weight.grad).abs().max().item() < TOL * gs


def test_rows16_full_width_workgroups(device, emulating):
    """conv3x3_rows16_fwd_kernel<64> (a workgroup = 64 output channels) runs from three workgroups per CU on; the test shapes and
    MedT's own (bs 4) take the half-width <32> instance.  MEDT_R16_OT=64 forces the full-width one onto the same parity cases."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MEDT_R16_OT="64")
    sel = "test_conv_block and (64-128-3-1-1 or 128-64-3-1-1-1-0 or 48-80)" if not emulating else "test_conv_block and train and 48-80"
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider",
                        os.path.join(root, "tests", "test_ops_gpu.py"), "-k", sel] + (["--emulate"] if emulating else []),
                       env=env, capture_output=True, text=True, timeout=7000 if emulating else 900, cwd=root)
    assert r.returncode == 0 and " passed" in r.stdout, r.stdout[-1500:] + r.stderr[-500:]
