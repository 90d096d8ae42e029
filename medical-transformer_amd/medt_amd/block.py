"""Host side of the one-launch AxialBlock_wopos forward (include/medt_abi.h: medt_wopos_block_fwd; csrc/block_small.hip).

The deep blocks of MedT's local branch (reference lib/models/axialnet.py:368-391 on the 4x4 maps of layer3_p) ran
conv_down+bn1+ReLU, two attention layers and conv_up+bn2+identity+ReLU as four dependent launches.  `fused_forward` runs
the whole block as one launch that writes exactly the tensors the four stages save for their backward; the four
autograd Functions are then applied in "adopt" mode (`pre=`): they wrap the precomputed outputs, save what they always
save, and launch nothing -- the autograd graph, and with it the whole backward, is the one of the per-stage path.

`block_forward` (default; MEDT_BLOCK_BWD=0 disables) wraps the same forward launch into ONE autograd Function whose backward is the
one-launch block backward (medt_wopos_block_bwd): six dependent launches -> one.  That kernel is verified against the
reference fixture on the CPU lane emulator (tests/test_lane_emu.py) and, since round 5, on the MI355X (tests/test_block_gpu.py);
it is ON by default.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib as L
from . import defer as DEFER
from .axial import _bn_ptrs

ENABLED = os.environ.get("MEDT_BLOCK_FUSED", "1") != "0" and os.environ.get("MEDT_DISABLE_SMALL", "0") != "1"
BWD_ENABLED = ENABLED and os.environ.get("MEDT_BLOCK_BWD", "1") != "0"     # (the library reads the same variable)


def _axial_params(att, training) -> L.AxialParams:
    return L.AxialParams(L.ptr(att.qkv_transform.weight), _bn_ptrs(att.bn_qkv, training), _bn_ptrs(att.bn_similarity, training),
                         _bn_ptrs(att.bn_output, training), None, None, None, None, None)


def fused_forward(blk, x, bn_groups: int):
    """None when the block / shape is not one the fused kernel is built for; else the precomputed stage outputs
    {"down": (z1, y1, stats1), "h": (qkv_raw, stacked, lse, stats, y_h), "w": (...), "up": (z2, y, stats2)}."""
    from . import ops
    if not ENABLED or ops.lean() or not x.is_cuda or x.dtype != torch.float32:
        return None
    if blk.downsample is not None:
        return fused_forward_s2(blk, x, bn_groups)
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
    # geometry the C side cannot see (it gets N, C, width, H, W, G only): the reference block's and nothing else
    for cv, cin, cout in ((blk.conv_down, Cc, width), (blk.conv_up, width, Cc)):
        if (tuple(cv.kernel_size) != (1, 1) or tuple(cv.stride) != (1, 1) or tuple(cv.padding) != (0, 0) or cv.groups != 1
                or cv.in_channels != cin or cv.out_channels != cout):
            return None
    if h.width or not w.width or h.groups != w.groups or h.training != training or w.training != training or blk.training != training:
        return None
    if h.qkv_transform.weight.shape[:2] != (2 * width, width) or w.qkv_transform.weight.shape[:2] != (2 * width, width):
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


def fused_forward_s2(blk, x, bn_groups: int):
    """The stride-2 FIRST block of a layer with its downsample path (reference lib/models/axialnet.py:368-391 + :596-606; layer4_p.0
    of MedT at 128 px) as one launch (medt_wopos_block_s2_fwd, round 6): the same dictionary as fused_forward plus
    "ds": (zd, yd, statsd) for the downsample conv block; the width layer's y is the pooled (N, width, H/2, W/2) tensor."""
    h, w, ds = blk.hight_block, blk.width_block, blk.downsample
    if h._has_pos or w._has_pos or h._gate_mode or w._gate_mode or h.stride != 1 or w.stride != 2:
        return None
    if len(ds) != 2 or not isinstance(ds[0], torch.nn.Conv2d) or not isinstance(ds[1], torch.nn.BatchNorm2d):
        return None
    bns = (blk.bn1, h.bn_qkv, h.bn_similarity, h.bn_output, w.bn_qkv, w.bn_similarity, w.bn_output, blk.bn2, ds[1])
    training = blk.bn1.training
    if any(b.training != training or b.momentum is None or b.eps != blk.bn1.eps or b.momentum != blk.bn1.momentum
           or b.running_mean is None for b in bns):
        return None
    N, Cc, H, W = x.shape
    width = blk.conv_down.weight.shape[0]
    Co = blk.conv_up.weight.shape[0]
    if Co != 2 * width or H % 2 or W % 2:
        return None
    for cv, cin, cout, st in ((blk.conv_down, Cc, width, 1), (blk.conv_up, width, Co, 1), (ds[0], Cc, Co, 2)):
        if (tuple(cv.kernel_size) != (1, 1) or tuple(cv.stride) != (st, st) or tuple(cv.padding) != (0, 0) or cv.groups != 1
                or cv.in_channels != cin or cv.out_channels != cout or cv.bias is not None):
            return None
    if h.width or not w.width or h.groups != w.groups or h.training != training or w.training != training or blk.training != training:
        return None
    if h.qkv_transform.weight.shape[:2] != (2 * width, width) or w.qkv_transform.weight.shape[:2] != (2 * width, width):
        return None
    lib = L.lib()
    desc = L.BlockDesc(N, Cc, width, H, W, h.groups, int(training), bn_groups, blk.bn1.eps, float(blk.bn1.momentum))
    ws_bytes = lib.medt_wopos_block_s2_workspace_bytes(C.byref(desc))
    if ws_bytes == 0:
        return None
    x = x.contiguous()
    dev = x.device
    f32 = dict(device=dev, dtype=torch.float32)
    G, H2, W2 = h.groups, H // 2, W // 2
    z1 = torch.empty((N, width, H, W), **f32)
    y1 = torch.empty_like(z1)
    stats1 = torch.empty((4 * bn_groups * width,), **f32)
    sv = []
    for l in range(2):
        sv.append((torch.empty((N, 2 * width, H, W), **f32), torch.empty((N, width, H, W), **f32),
                   torch.empty((N, G, H, W), **f32), torch.empty((4 * bn_groups * (2 * width + G + width),), **f32),
                   torch.empty((N, width, H, W) if l == 0 else (N, width, H2, W2), **f32)))
    z2 = torch.empty((N, Co, H2, W2), **f32)
    y = torch.empty_like(z2)
    stats2 = torch.empty((4 * bn_groups * Co,), **f32)
    zd = torch.empty_like(z2)
    yd = torch.empty_like(z2)
    statsd = torch.empty((4 * bn_groups * Co,), **f32)
    ws = torch.empty((ws_bytes,), device=dev, dtype=torch.uint8)
    params = L.BlockS2Params(L.BlockParams(L.ptr(blk.conv_down.weight), _bn_ptrs(blk.bn1, training), _axial_params(h, training),
                                           _axial_params(w, training), L.ptr(blk.conv_up.weight), _bn_ptrs(blk.bn2, training)),
                             L.ptr(ds[0].weight), _bn_ptrs(ds[1], training))
    saved = L.BlockS2Saved(L.BlockSaved(z1.data_ptr(), y1.data_ptr(), stats1.data_ptr(),
                                        L.AxialSaved(sv[0][0].data_ptr(), sv[0][1].data_ptr(), sv[0][2].data_ptr(), sv[0][3].data_ptr()),
                                        sv[0][4].data_ptr(),
                                        L.AxialSaved(sv[1][0].data_ptr(), sv[1][1].data_ptr(), sv[1][2].data_ptr(), sv[1][3].data_ptr()),
                                        sv[1][4].data_ptr(), z2.data_ptr(), stats2.data_ptr()),
                           zd.data_ptr(), yd.data_ptr(), statsd.data_ptr())
    q = DEFER.recording()
    L.check(lib.medt_wopos_block_s2_fwd(C.byref(desc), C.byref(params), x.data_ptr(), y.data_ptr(), C.byref(saved),
                                        ws.data_ptr(), ws_bytes, torch.cuda.current_stream().cuda_stream), "medt_wopos_block_s2_fwd")
    if q is not None:
        q.hold(ws, stats1, sv[0][3], sv[1][3], stats2, statsd)
    return {"down": (z1, y1, stats1), "h": sv[0], "w": sv[1], "up": (z2, y, stats2), "ds": (zd, yd, statsd)}


# --------------------------------------------------------------------------- #
# the whole block as one autograd node: one-launch forward, one-launch backward
# --------------------------------------------------------------------------- #
def _block_params(blk):
    """The 20 trained tensors of an AxialBlock_wopos in gradient order (conv1 is registered and never used)."""
    h, w = blk.hight_block, blk.width_block
    out = [blk.conv_down.weight, blk.bn1.weight, blk.bn1.bias]
    for a in (h, w):
        out += [a.qkv_transform.weight, a.bn_qkv.weight, a.bn_qkv.bias, a.bn_similarity.weight, a.bn_similarity.bias,
                a.bn_output.weight, a.bn_output.bias]
    return out + [blk.conv_up.weight, blk.bn2.weight, blk.bn2.bias]


class WoposBlockFn(torch.autograd.Function):
    """y = AxialBlock_wopos(x) (reference lib/models/axialnet.py:368-391).  Inputs: x, then the 20 tensors of _block_params."""

    @staticmethod
    def forward(ctx, x, *rest):
        from . import optim as OPT
        params, (blk, bn_groups, sink, pre) = rest[:20], rest[20:]
        ctx.geom = (x.shape[0], x.shape[1], params[0].shape[0], x.shape[2], x.shape[3], blk.hight_block.groups,
                    int(blk.bn1.training), bn_groups, blk.bn1.eps, float(blk.bn1.momentum))
        ctx.sink = sink
        ctx.slots = tuple(OPT.grad_slot(t) for t in params)
        z1, y1, stats1 = pre["down"]
        z2, y, stats2 = pre["up"]
        ctx.save_for_backward(x, y, z1, y1, stats1, *pre["h"], *pre["w"], z2, stats2, *params)
        return y

    @staticmethod
    def backward(ctx, dy):
        from . import optim as OPT
        lib = L.lib()
        sv = ctx.saved_tensors
        x, y, z1, y1, stats1 = sv[:5]
        qh, sh, lh, sth, yh = sv[5:10]
        qw, sw, lw, stw, yw = sv[10:15]
        z2, stats2 = sv[15:17]
        P = sv[17:]
        dy = dy.contiguous()
        dev = x.device
        desc = L.BlockDesc(*ctx.geom)
        ws_bytes = lib.medt_wopos_block_bwd_workspace_bytes(C.byref(desc))
        if ws_bytes == 0:
            raise L.MedtError("block backward: " + (lib.medt_last_error().decode() or "MEDT_BLOCK_BWD changed since the forward"))
        # gradient destinations: the parameter's slot in FlatAdam's flat bucket, else a temporary handed back to autograd
        dst, ret, pend = [None] * 20, [None] * 20, []
        for k in range(20):
            slot = OPT.live(ctx.slots[k])
            if slot is not None and ctx.needs_input_grad[k + 1]:
                dst[k], direct = OPT.claim(slot)
                if not direct:
                    pend.append((slot, dst[k]))
            else:
                dst[k] = torch.empty(P[k].shape, device=dev, dtype=torch.float32)
                if ctx.needs_input_grad[k + 1]:
                    ret[k] = dst[k]
        g = [L.ptr(t) for t in dst]
        grads = L.BlockGrads(g[0], g[1], g[2], L.AxialGrads(*g[3:10], None, None), L.AxialGrads(*g[10:17], None, None),
                             g[17], g[18], g[19])
        bn = lambda w: L.BnPtrs(L.ptr(w), None, None, None, None)          # the backward reads the BatchNorm weights only
        ax = lambda o: L.AxialParams(L.ptr(P[o]), bn(P[o + 1]), bn(P[o + 3]), bn(P[o + 5]), None, None, None, None, None)
        params = L.BlockParams(L.ptr(P[0]), bn(P[1]), ax(3), ax(10), L.ptr(P[17]), bn(P[18]))
        saved = L.BlockSaved(z1.data_ptr(), y1.data_ptr(), stats1.data_ptr(),
                             L.AxialSaved(qh.data_ptr(), sh.data_ptr(), lh.data_ptr(), sth.data_ptr()), yh.data_ptr(),
                             L.AxialSaved(qw.data_ptr(), sw.data_ptr(), lw.data_ptr(), stw.data_ptr()), yw.data_ptr(),
                             z2.data_ptr(), stats2.data_ptr())
        dx = torch.empty(x.shape, device=dev, dtype=torch.float32)
        # fan-in of d(x): this node is conv_down AND the identity path; what other consumers deposited is added in the kernel
        xs = ctx.sink if ctx.needs_input_grad[0] else None
        dep = xs.take() if xs is not None else None
        ws = torch.empty((ws_bytes,), device=dev, dtype=torch.uint8)
        q = DEFER.recording(allow=not pend and all(r is None for r in ret))
        L.check(lib.medt_wopos_block_bwd(C.byref(desc), C.byref(params), x.data_ptr(), y.data_ptr(), dy.data_ptr(), C.byref(saved),
                                         dx.data_ptr(), L.ptr(dep), C.byref(grads), ws.data_ptr(), ws_bytes,
                                         torch.cuda.current_stream().cuda_stream), "medt_wopos_block_bwd")
        if q is not None:                      # the recorded weight-gradient / reduction jobs read these at the flush
            q.hold(ws, x, y1, yh, yw, qh, qw, stats1, sth, stw, stats2, *dst)
        for slot, tmp in pend:
            OPT.accumulate(slot, tmp)
        if xs is not None:
            xs.closed = True                   # role "final": later depositors return their gradient the ordinary way
        return (dx if ctx.needs_input_grad[0] else None, *ret, None, None, None, None)


def block_forward(blk, x, bn_groups: int, sink):
    """The block as ONE autograd node (one-launch forward and backward), or None when that path is not taken."""
    if not BWD_ENABLED or not torch.is_grad_enabled() or blk.downsample is not None:
        return None
    # (the one-launch backward exists for fewer shapes than the forward: ask before committing this block to the one-node path)
    N, Cc, H, W = x.shape
    desc = L.BlockDesc(N, Cc, blk.conv_down.weight.shape[0], H, W, blk.hight_block.groups, int(blk.bn1.training), bn_groups,
                       blk.bn1.eps, float(blk.bn1.momentum if blk.bn1.momentum is not None else 0.1))
    if L.lib().medt_wopos_block_bwd_workspace_bytes(C.byref(desc)) == 0:
        return None
    x = x.contiguous()                     # the backward reads x (and writes dx) as dense NCHW through raw pointers
    pre = fused_forward(blk, x, bn_groups)
    if pre is None:
        return None
    return WoposBlockFn.apply(x, *_block_params(blk), blk, bn_groups, sink, pre)
