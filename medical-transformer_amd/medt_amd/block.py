"""Host side of the one-launch AxialBlock_wopos forward (include/medt_abi.h: medt_wopos_block_fwd; csrc/block_small.hip).

The deep blocks of MedT's local branch (reference lib/models/axialnet.py:368-391 on the 4x4 maps of layer3_p) ran
conv_down+bn1+ReLU, two attention layers and conv_up+bn2+identity+ReLU as four dependent launches.  `fused_forward` runs
the whole block as one launch that writes exactly the tensors the four stages save for their backward; the four
autograd Functions are then applied in "adopt" mode (`pre=`): they wrap the precomputed outputs, save what they always
save, and launch nothing -- the autograd graph, and with it the whole backward, is the one of the per-stage path.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib as L
from . import defer as DEFER
from .axial import _bn_ptrs

ENABLED = os.environ.get("MEDT_BLOCK_FUSED", "1") != "0" and os.environ.get("MEDT_DISABLE_SMALL", "0") != "1"


def _axial_params(att, training) -> L.AxialParams:
    return L.AxialParams(L.ptr(att.qkv_transform.weight), _bn_ptrs(att.bn_qkv, training), _bn_ptrs(att.bn_similarity, training),
                         _bn_ptrs(att.bn_output, training), None, None, None, None, None)


def fused_forward(blk, x, bn_groups: int):
    """None when the block / shape is not one the fused kernel is built for; else the precomputed stage outputs
    {"down": (z1, y1, stats1), "h": (qkv_raw, stacked, lse, stats, y_h), "w": (...), "up": (z2, y, stats2)}."""
    from . import ops
    if not ENABLED or ops.LEAN or not x.is_cuda or x.dtype != torch.float32 or blk.downsample is not None:
        return None
    h, w = blk.hight_block, blk.width_block
    if h._has_pos or w._has_pos or h._gate_mode or w._gate_mode or h.stride != 1 or w.stride != 1:
        return None
    bns = (blk.bn1, h.bn_qkv, h.bn_similarity, h.bn_output, w.bn_qkv, w.bn_similarity, w.bn_output, blk.bn2)
    training = blk.bn1.training
    if any(b.training != training or b.momentum is None or b.eps != blk.bn1.eps or b.momentum != blk.bn1.momentum
           or b.running_mean is None for b in bns):
        return None
    if blk.conv_down.bias is not None or blk.conv_up.bias is not None:
        return None
    N, Cc, H, W = x.shape
    width = blk.conv_down.weight.shape[0]
    if blk.conv_up.weight.shape[0] != Cc:
        return None
    lib = L.lib()
    desc = L.BlockDesc(N, Cc, width, H, W, h.groups, int(training), bn_groups, blk.bn1.eps, float(blk.bn1.momentum))
    ws_bytes = lib.medt_wopos_block_workspace_bytes(C.byref(desc))
    if ws_bytes == 0:
        return None
    x = x.contiguous()
    dev = x.device
    f32 = dict(device=dev, dtype=torch.float32)
    G = h.groups
    z1 = torch.empty((N, width, H, W), **f32)
    y1 = torch.empty_like(z1)
    stats1 = torch.empty((4 * bn_groups * width,), **f32)
    sv = []
    for _ in range(2):
        sv.append((torch.empty((N, 2 * width, H, W), **f32), torch.empty((N, width, H, W), **f32),
                   torch.empty((N, G, H, W), **f32), torch.empty((4 * bn_groups * (2 * width + G + width),), **f32),
                   torch.empty((N, width, H, W), **f32)))
    z2 = torch.empty((N, Cc, H, W), **f32)
    y = torch.empty_like(z2)
    stats2 = torch.empty((4 * bn_groups * Cc,), **f32)
    ws = torch.empty((ws_bytes,), device=dev, dtype=torch.uint8)
    params = L.BlockParams(L.ptr(blk.conv_down.weight), _bn_ptrs(blk.bn1, training), _axial_params(h, training),
                           _axial_params(w, training), L.ptr(blk.conv_up.weight), _bn_ptrs(blk.bn2, training))
    saved = L.BlockSaved(z1.data_ptr(), y1.data_ptr(), stats1.data_ptr(),
                         L.AxialSaved(sv[0][0].data_ptr(), sv[0][1].data_ptr(), sv[0][2].data_ptr(), sv[0][3].data_ptr()),
                         sv[0][4].data_ptr(),
                         L.AxialSaved(sv[1][0].data_ptr(), sv[1][1].data_ptr(), sv[1][2].data_ptr(), sv[1][3].data_ptr()),
                         sv[1][4].data_ptr(), z2.data_ptr(), stats2.data_ptr())
    q = DEFER.recording()
    L.check(lib.medt_wopos_block_fwd(C.byref(desc), C.byref(params), x.data_ptr(), y.data_ptr(), C.byref(saved),
                                     ws.data_ptr(), ws_bytes, torch.cuda.current_stream().cuda_stream), "medt_wopos_block_fwd")
    if q is not None:                      # the recorded statistics jobs read the partial sums and write the stats blocks
        q.hold(ws, stats1, sv[0][3], sv[1][3], stats2)
    return {"down": (z1, y1, stats1), "h": sv[0], "w": sv[1], "up": (z2, y, stats2)}
