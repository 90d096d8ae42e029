"""Network-level execution for the lib.models.axialnet module surface.

Walks the same dataflow as the reference's block / network forwards
(lib/models/axialnet.py:324-344, 471-504, 620-708); the modules in
lib/models/axialnet.py only hold parameters and delegate here.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

from ._lib import MedtError


def _require_device(x):
    if not x.is_cuda:
        raise MedtError("medt_amd runs on MI355X only: got a CPU tensor and there is no CPU fallback "
                        "(the CPU restatement under oracle/ is test infrastructure)")


def axial_block_forward(blk, x):
    """AxialBlock{,_dynamic,_wopos}.forward (reference :282-302, :324-344, :368-391)."""
    _require_device(x)
    out = F.relu(blk.bn1(blk.conv_down(x)))
    out = blk.hight_block(out)
    out = blk.width_block(out)
    out = F.relu(out)
    out = blk.bn2(blk.conv_up(out))
    identity = x if blk.downsample is None else blk.downsample(x)
    return F.relu(out + identity)


def _up(x):
    return F.interpolate(x, scale_factor=(2, 2), mode="bilinear", align_corners=False)


def _stem(net, x, sfx=""):
    g = lambda n: getattr(net, n + sfx)
    x = F.relu(g("bn1")(g("conv1")(x)))
    x = F.relu(g("bn2")(g("conv2")(x)))
    x = F.relu(g("bn3")(g("conv3")(x)))
    return x


def _unet_body(net, x, sfx=""):
    g = lambda n: getattr(net, n + sfx)
    x1 = g("layer1")(x)
    x2 = g("layer2")(x1)
    x3 = g("layer3")(x2)
    x4 = g("layer4")(x3)
    y = F.relu(_up(g("decoder1")(x4))) + x4
    y = F.relu(_up(g("decoder2")(y))) + x3
    y = F.relu(_up(g("decoder3")(y))) + x2
    y = F.relu(_up(g("decoder4")(y))) + x1
    y = F.relu(_up(g("decoder5")(y)))
    return y


def unet_forward(net, x):
    """ResAxialAttentionUNet._forward_impl (reference :471-504)."""
    _require_device(x)
    y = _unet_body(net, _stem(net, x))
    return net.adjust(F.relu(y))


def medt_forward(net, x):
    """medt_net._forward_impl (reference :620-708)."""
    _require_device(x)
    xin = x
    g = _stem(net, x)
    x1 = net.layer1(g)
    x2 = net.layer2(x1)
    y = F.relu(_up(net.decoder4(x2))) + x1
    y = F.relu(_up(net.decoder5(y)))
    x_loc = y.clone()
    for i in range(4):                       # hard-coded 4x4 grid of 32-px patches (:661-664)
        for j in range(4):
            xp = xin[:, :, 32 * i:32 * i + 32, 32 * j:32 * j + 32]
            yp = _unet_body(net, _stem(net, xp, "_p"), "_p")
            x_loc[:, :, 32 * i:32 * i + 32, 32 * j:32 * j + 32] = yp
    y = y + x_loc
    y = F.relu(net.decoderf(y))
    return net.adjust(F.relu(y))
