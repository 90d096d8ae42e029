"""Network-level execution for the lib.models.axialnet module surface.

Walks the same dataflow as the reference's block / network forwards
(lib/models/axialnet.py:324-344, 471-504, 620-708) with every op bound to
libmedt_hip.so; the modules in lib/models/axialnet.py only hold parameters and
delegate here.  MI355X-first differences in *schedule*, not in arithmetic:

  * conv + BatchNorm + residual + ReLU run as one fused block op, bilinear-x2 + ReLU + skip as one kernel,
    the ReLU after each block's width attention is fused into the attention's output pass;
  * the reference's 16-iteration Python patch loop (:661-700) becomes ONE pass over a patch-major
    (16N, C, 32, 32) stack in which every BatchNorm keeps 16 statistic groups and applies the
    running-stat recurrence in patch order -- the same numbers, 16x fewer launches.
"""
from __future__ import annotations

import torch

from ._lib import MedtError
from . import ops
from . import defer as DEFER
from . import block as BLOCK

import os

PATCH = 32          # hard-coded in the reference (:664)
GRID = 4
TWO_STREAMS = os.environ.get("MEDT_TWO_STREAMS", "1") != "0"
SINKS = os.environ.get("MEDT_GRAD_SINKS", "1") != "0"       # gradient fan-in in dgrad epilogues (ops.GradSink)
EARLY_FIN = os.environ.get("MEDT_EARLY_FIN", "1") != "0"    # forward bookkeeping flushed per branch, off the loss trunk
_side = {}


def _side_stream(device):
    key = (device.type, device.index)
    if key not in _side:
        _side[key] = torch.cuda.Stream(device=device)
    return _side[key]


def join_side_stream(device):
    """Make the current stream wait for everything issued on the branch stream so far.  A forward WITHOUT a backward behind it
    (trainer.InferStep in train mode) leaves the local branch's recorded bookkeeping -- issued on its stream behind the event the
    merge waits for -- unjoined; a stream capture must not end like that."""
    side = _side.get((device.type, device.index))
    if side is not None:
        torch.cuda.current_stream().wait_stream(side)


def _require_device(x):
    if not x.is_cuda:
        raise MedtError("medt_amd runs on MI355X only: got a CPU tensor and there is no CPU fallback "
                        "(the CPU restatement under oracle/ is test infrastructure)")


def axial_block_forward(blk, x, bn_groups: int = 1):
    """AxialBlock{,_dynamic,_wopos}.forward (reference :282-302, :324-344, :368-391)."""
    _require_device(x)
    # x fans out to conv_down and to the residual / downsample path (and, for layer outputs, to a decoder skip): their
    # gradients meet in conv_down's dgrad epilogue instead of autograd add kernels (ops.GradSink)
    sink = ops.sink_of(x) if (SINKS and x.requires_grad) else None
    # the deep position-free blocks of the local branch: the whole block forward is ONE launch (block.py); the four stages
    # below then adopt its outputs (`pre`) instead of launching -- same autograd graph, same backward
    if BLOCK.BWD_ENABLED:                  # ... and (default; MEDT_BLOCK_BWD=0 disables) its backward too: the block is one autograd node
        y = BLOCK.block_forward(blk, x, bn_groups, sink)
        if y is not None:
            return y
    pre = BLOCK.fused_forward(blk, x, bn_groups)
    if pre is None:
        pre = {"down": None, "h": None, "w": None, "up": None}
    pre.setdefault("ds", None)
    out = ops.conv_block(x, blk.conv_down, blk.bn1, relu=True, training=blk.bn1.training, bn_groups=bn_groups,
                         x_sink=sink, x_role="final", pre=pre["down"])
    out = blk.hight_block.run(out, bn_groups, False, pre["h"])
    out = blk.width_block.run(out, bn_groups, True, pre["w"])    # + the block's ReLU (:333)
    if blk.downsample is not None:
        identity = ops.conv_block(x, blk.downsample[0], blk.downsample[1], relu=False,
                                  training=blk.downsample[1].training, bn_groups=bn_groups, x_sink=sink, x_role="deposit",
                                  pre=pre["ds"])
        return ops.conv_block(out, blk.conv_up, blk.bn2, res=identity, relu=True, training=blk.bn2.training,
                              bn_groups=bn_groups, pre=pre["up"])
    return ops.conv_block(out, blk.conv_up, blk.bn2, res=x, relu=True, training=blk.bn2.training,
                          bn_groups=bn_groups, res_sink=sink, pre=pre["up"])


def _layer(seq, x, bn_groups=1):
    for blk in seq:
        x = axial_block_forward(blk, x, bn_groups)
    return x


def _stem(net, x, sfx="", bn_groups=1, first_of_branch=False):
    g = lambda n: getattr(net, n + sfx)
    for i in ("1", "2", "3"):
        bn = g("bn" + i)
        x = ops.conv_block(x, g("conv" + i), bn, relu=True, training=bn.training, bn_groups=bn_groups,
                           last_of_branch=first_of_branch and i == "1")     # first forward op = last backward op
    return x


def _unet_body(net, x, sfx="", bn_groups=1):
    g = lambda n: getattr(net, n + sfx)
    x1 = _layer(g("layer1"), x, bn_groups)
    x2 = _layer(g("layer2"), x1, bn_groups)
    x3 = _layer(g("layer3"), x2, bn_groups)
    x4 = _layer(g("layer4"), x3, bn_groups)
    sk = (lambda t: ops.sink_of(t)) if SINKS else (lambda t: None)    # x1..x3 also feed the next layer's first block
    y = ops.up2x_relu_add(ops.conv_block(x4, g("decoder1")), x4)
    y = ops.up2x_relu_add(ops.conv_block(y, g("decoder2")), x3, sk(x3))
    y = ops.up2x_relu_add(ops.conv_block(y, g("decoder3")), x2, sk(x2))
    y = ops.up2x_relu_add(ops.conv_block(y, g("decoder4")), x1, sk(x1))
    y = ops.up2x_relu_add(ops.conv_block(y, g("decoder5")), None)
    return y


def unet_forward(net, x):
    """ResAxialAttentionUNet._forward_impl (reference :471-504).  adjust(relu(y)) == adjust(y): y is a ReLU output."""
    _require_device(x)
    y = _unet_body(net, _stem(net, x.contiguous()))
    return ops.conv_block(y, net.adjust)


def medt_forward(net, x):
    """medt_net._forward_impl (reference :620-708)."""
    _require_device(x)
    xin = x.contiguous()
    if xin.shape[2] < PATCH * GRID or xin.shape[3] < PATCH * GRID or xin.shape[2] != xin.shape[3]:
        raise RuntimeError(f"medt_net needs square images of at least {PATCH * GRID} px (4x4 grid of 32-px patches)")
    # The global and the local (patch) branch only meet at the merge: run them on two HIP streams so the
    # latency-bound small kernels of one overlap the other's (both forks are captured into the step's hipGraph;
    # autograd replays the backward of each branch on the stream its forward ran on).
    main = torch.cuda.current_stream()
    side = _side_stream(xin.device) if TWO_STREAMS else None
    if side is not None:
        side.wait_stream(main)
    DEFER.set_aux_stream(side)             # (the pass's last flush forks its MFMA weight gradients onto the idle branch stream)
    # (the global branch's backward ends well before the local one's: its recorded weight-gradient jobs are issued on
    # its own stream as soon as its stem's backward has run, under the local chain -- defer.flush_current_stream)
    g = _stem(net, xin, first_of_branch=side is not None)
    x1 = _layer(net.layer1, g)
    x2 = _layer(net.layer2, x1)
    y = ops.up2x_relu_add(ops.conv_block(x2, net.decoder4), x1, ops.sink_of(x1) if SINKS else None)
    y = ops.up2x_relu_add(ops.conv_block(y, net.decoder5), None)
    if side is not None and EARLY_FIN:
        # the global branch's forward ends here, ~100 us before the local one's: its recorded bookkeeping (running
        # statistics) is issued now, while this stream would wait for the other, instead of after the loss
        DEFER.flush_current_stream()
    # local branch: all 16 patches at once, patch-major on the batch dim, one BatchNorm group per patch.  In eval mode
    # the grouping does not change the result (running statistics); it is kept so the small per-group slices still
    # take the fused small-layer kernels (2 launches per layer instead of 6)
    groups = GRID * GRID
    # beyond 128 px the global branch dominates the step and runs CU-filling persistent attention kernels (L = 128): the
    # local branch then keeps to kernels whose workgroups co-reside with them (ops.set_lean -> medt_conv_desc.lean; a per-thread hint)
    ops.set_lean(side is not None and xin.shape[2] * xin.shape[3] > 128 * 128)
    try:
        if side is not None:
            with torch.cuda.stream(side):
                xp = ops.patch_gather(xin, PATCH, GRID)
                yp = _unet_body(net, _stem(net, xp, "_p", groups), "_p", groups)
                if EARLY_FIN:
                    # the local branch's recorded bookkeeping (saved statistics of the fused small layers, running
                    # statistics) is issued on ITS stream behind an event the merge waits for: it runs under the merge /
                    # decoderf / loss kernels of the main stream and still precedes the local backward, which is ordered
                    # on this stream
                    joined = torch.cuda.Event()
                    joined.record(side)
                    DEFER.flush_current_stream()
            if EARLY_FIN:
                main.wait_event(joined)
            else:
                main.wait_stream(side)
            yp.record_stream(main)
        else:
            xp = ops.patch_gather(xin, PATCH, GRID)
            yp = _unet_body(net, _stem(net, xp, "_p", groups), "_p", groups)
    finally:
        ops.set_lean(False)            # (the hint is read when a block's configuration is built: forward time)
    y = ops.logo_merge(y, yp, PATCH, GRID)
    y = ops.conv_block(y, net.decoderf, relu=True)
    return ops.conv_block(y, net.adjust)
