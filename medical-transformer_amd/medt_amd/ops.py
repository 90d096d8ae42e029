"""autograd.Functions over the C ABI for everything around the attention layers:
convolution blocks (conv + BatchNorm + residual + ReLU), bilinear-x2 + ReLU + skip,
the LoGo patch gather / merge and the cross-entropy loss.

Reference lines: lib/models/axialnet.py:285-300, 450-454, 475-502, 623-705; metrics.py:17-20.
PyTorch supplies device memory, the stream and the autograd graph only.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib as L
from . import optim as OPT
from . import defer as DEFER
from .axial import _bn_ptrs, _momentum, _require_device


def _stream():
    return torch.cuda.current_stream().cuda_stream


# --------------------------------------------------------------------------- #
# y = act( BN( conv(x) + bias ) + res )
# --------------------------------------------------------------------------- #
class GradSink:
    """Gradient fan-in without add kernels.  A block input x feeds conv_down AND the residual / downsample path, the layer
    outputs x1..x3 also feed a decoder skip: the reference's autograd sums those gradients with one add kernel per extra
    consumer.  Consumers that share a sink hand their contribution over instead: whoever finishes its backward first
    deposits its gradient tensor here and returns nothing to autograd; the next one adds the deposit in the epilogue of
    its own dgrad kernel (medt_conv_block_bwd's dx_add) and either re-deposits the sum (role "deposit") or -- the
    consumer autograd runs last, conv_down: the earliest-created node -- returns it (role "final").  If the order ever
    differs (final already ran) a depositor simply returns its gradient the ordinary way."""
    __slots__ = ("pending", "closed")

    def __init__(self):
        self.pending, self.closed = None, False

    def deposit(self, t) -> bool:
        """True: taken (the caller returns None to autograd).  False: return the gradient normally."""
        if self.closed or self.pending is not None:
            return False
        self.pending = t
        return True

    def take(self):
        t, self.pending = self.pending, None
        return t


def sink_of(x):
    """The sink shared by the consumers of tensor x (created on first use)."""
    s = getattr(x, "_medt_sink", None)
    if s is None:
        s = x._medt_sink = GradSink()
    return s


# Scheduling hint (net.medt_forward sets it per call): True when the OTHER branch's stream runs CU-filling persistent kernels
# (MedT's global branch beyond 128 px): the local branch then keeps to kernels with small LDS footprints (medt_conv_desc.lean)
# Per THREAD (nn.DataParallel's thread-per-replica model, a validation thread next to a training thread): a forward of one thread
# must not change the kernels another thread's forward picks.
import threading
_hint = threading.local()


def set_lean(v: bool):
    _hint.lean = bool(v)


def lean() -> bool:
    return getattr(_hint, "lean", False)



class ConvBlockCfg:
    __slots__ = ("stride", "pad", "bn", "relu", "bn_groups", "x_sink", "x_role", "res_sink", "last_of_branch", "pre", "lean")

    def __init__(self, stride, pad, bn, relu, bn_groups=1, x_sink=None, x_role=None, res_sink=None, last_of_branch=False,
                 pre=None):
        self.stride, self.pad, self.bn, self.relu, self.bn_groups = stride, pad, bn, relu, bn_groups
        self.x_sink, self.x_role, self.res_sink = x_sink, x_role, res_sink
        self.last_of_branch = last_of_branch       # this block's backward is the last work of its stream's backward pass
        self.pre = pre                             # (z, y, stats) already computed by the one-launch block forward (block.py)
        self.lean = lean()


def _conv_desc(x, w, cfg: ConvBlockCfg, has_bias, has_res, training) -> L.ConvDesc:
    N, Cin, H, W = x.shape
    bn = cfg.bn
    return L.ConvDesc(N, Cin, H, W, w.shape[0], w.shape[2], cfg.stride, cfg.pad, int(has_bias), int(bn is not None),
                      int(has_res), int(cfg.relu), int(training), cfg.bn_groups,
                      bn.eps if bn is not None else 1e-5,
                      _momentum(bn) if bn is not None else 0.1, int(cfg.lean))


class ConvBlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, bn_w, bn_b, res, cfg: ConvBlockCfg, training: bool):
        _require_device(x)
        lib = L.lib()
        x = x.contiguous()
        if res is not None:
            res = res.contiguous()
        desc = _conv_desc(x, w, cfg, bias is not None, res is not None, training)
        K, s, p = w.shape[2], cfg.stride, cfg.pad
        N, _, H, W = x.shape
        Ho, Wo = (H + 2 * p - K) // s + 1, (W + 2 * p - K) // s + 1
        has_bn = cfg.bn is not None
        if cfg.pre is not None:                    # adopt mode: the one-launch block forward already produced these
            z, y, stats = cfg.pre
            cfg.pre = None
        else:
            y = torch.empty((N, w.shape[0], Ho, Wo), device=x.device, dtype=torch.float32)
            z = torch.empty_like(y) if has_bn else y
            stats = torch.empty((max(lib.medt_conv_stats_floats(C.byref(desc)), 1),), device=x.device, dtype=torch.float32)
            ws_bytes = lib.medt_conv_workspace_bytes(C.byref(desc))
            if ws_bytes == 0:
                raise L.MedtError("conv block: " + lib.medt_last_error().decode())
            ws = torch.empty((ws_bytes,), device=x.device, dtype=torch.uint8)
            bnp = _bn_ptrs(cfg.bn, training) if has_bn else None
            # (recorded jobs write into stats at the flush: the BatchNorm finalisation, and the weight flip of a training-mode
            #  3x3 layer with or without BatchNorm -- the queue keeps both buffers alive until then)
            q = DEFER.recording() if (has_bn or stats.numel() > 1) else None
            L.check(lib.medt_conv_block_fwd(C.byref(desc), x.data_ptr(), w.data_ptr(), L.ptr(bias),
                                            C.byref(bnp) if has_bn else None, L.ptr(res), z.data_ptr(), y.data_ptr(),
                                            stats.data_ptr(), ws.data_ptr(), ws_bytes, _stream()), "medt_conv_block_fwd")
            if q is not None:
                q.hold(ws, stats)
        ctx.cfg, ctx.training, ctx.has_bias, ctx.has_res = cfg, training, bias is not None, res is not None
        # gradient slots of (w, bias, bn.weight, bn.bias) in FlatAdam's flat bucket: backward writes them directly
        ctx.slots = tuple(OPT.grad_slot(t) if t is not None else None for t in (w, bias, bn_w, bn_b))
        # (stats: the BatchNorm statistics and, behind them, the flipped weights a training-mode forward leaves for the MFMA
        #  backward-data kernel -- medt_conv_stats_floats says how much; layers without either bring a one-float dummy)
        ctx.save_for_backward(x, w, z if has_bn else None, y if (cfg.relu or has_bn) else None,
                              stats if (has_bn or stats.numel() > 1) else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.lib()
        x, w, z, y, stats = ctx.saved_tensors
        cfg, training = ctx.cfg, ctx.training
        dy = dy.contiguous()
        desc = _conv_desc(x, w, cfg, ctx.has_bias, ctx.has_res, training)
        has_bn = cfg.bn is not None
        dev = x.device
        Cout = w.shape[0]
        need_dx = ctx.needs_input_grad[0]
        dx = torch.empty_like(x) if need_dx else None
        # fan-in of the input's gradient (GradSink): add what the other consumers deposited in this dgrad's epilogue
        xs = cfg.x_sink if need_dx else None
        dep = xs.take() if xs is not None else None
        dx_add = dep if (dep is not None and w.shape[2] == 1) else None       # the dgrad epilogue add exists for 1x1 only
        present = (True, ctx.has_bias, has_bn, has_bn)
        shapes = (w.shape, (Cout,), (Cout,), (Cout,))
        dst, ret, pend = [None] * 4, [None] * 4, []
        for k in range(4):
            if not present[k]:
                continue
            slot = OPT.live(ctx.slots[k])
            if slot is not None and ctx.needs_input_grad[k + 1]:
                dst[k], direct = OPT.claim(slot)
                if not direct:
                    pend.append((slot, dst[k]))
            else:
                dst[k] = torch.empty(shapes[k], device=dev, dtype=torch.float32)
                if ctx.needs_input_grad[k + 1]:
                    ret[k] = dst[k]
        dres = torch.empty_like(dy) if ctx.has_res else None
        ws_bytes = lib.medt_conv_workspace_bytes(C.byref(desc))
        ws = torch.empty((ws_bytes,), device=dev, dtype=torch.uint8)
        bnp = _bn_ptrs(cfg.bn, False) if has_bn else None
        # parameter gradients may only be recorded for the grouped flush when they all land in persistent slots
        q = DEFER.recording(allow=not pend and all(r is None for r in ret))
        L.check(lib.medt_conv_block_bwd(C.byref(desc), x.data_ptr(), w.data_ptr(), C.byref(bnp) if has_bn else None,
                                        L.ptr(z), L.ptr(y), L.ptr(stats), dy.data_ptr(), L.ptr(dx), dst[0].data_ptr(),
                                        L.ptr(dst[1]), L.ptr(dst[2]), L.ptr(dst[3]), L.ptr(dres), L.ptr(dx_add), ws.data_ptr(),
                                        ws_bytes, _stream()), "medt_conv_block_bwd")
        if q is not None:                      # recorded weight / bias gradient jobs read these at the flush
            q.hold(ws, x, dy, stats, dres, *dst)
        for slot, tmp in pend:
            OPT.accumulate(slot, tmp)
        if dep is not None and dx_add is None:         # a wider consumer: the deposit is added explicitly, never dropped
            dx.add_(dep)
        if xs is not None:
            if cfg.x_role == "final":
                xs.closed = True
            elif xs.deposit(dx):                       # role "deposit": hand dx to the consumer that runs after this one
                dx = None
        if dres is not None and cfg.res_sink is not None and cfg.res_sink.deposit(dres):
            dres = None
        if cfg.last_of_branch:                         # nothing else of this branch follows: its recorded jobs go out now
            if DEFER.flush_current_stream():
                OPT.branch_done(0)                     # ... and (data parallel) its gradient bucket goes on the wire
        return (dx, ret[0], ret[1], ret[2], ret[3], dres, None, None)


def conv_block(x, conv, bn=None, res=None, relu=False, training=False, bn_groups=1, x_sink=None, x_role=None,
               res_sink=None, last_of_branch=False, pre=None):
    """conv: nn.Conv2d holder, bn: nn.BatchNorm2d holder or None.  x_sink / x_role / res_sink: see GradSink.
    pre: (z, y, stats) computed by the one-launch block forward (medt_amd.block) -- nothing is launched then."""
    cfg = ConvBlockCfg(conv.stride[0], conv.padding[0], bn, relu, bn_groups if bn is not None else 1, x_sink, x_role,
                       res_sink, last_of_branch, pre)
    if bn is None:
        # without BatchNorm the descriptor's `training` flag only says "a backward pass follows": the forward then leaves the
        # flipped weights of the MFMA backward-data kernel behind (medt_conv_stats_floats), off the backward chain
        training = torch.is_grad_enabled() and (x.requires_grad or conv.weight.requires_grad)
    return ConvBlockFn.apply(x, conv.weight, conv.bias, bn.weight if bn is not None else None,
                             bn.bias if bn is not None else None, res, cfg, training)


# --------------------------------------------------------------------------- #
# y = relu(bilinear_x2(x)) + skip
# --------------------------------------------------------------------------- #
class UpReluAddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, skip, skip_sink=None):
        ctx.skip_sink = skip_sink
        _require_device(x)
        lib = L.lib()
        x = x.contiguous()
        N, Cc, H, W = x.shape
        y = torch.empty((N, Cc, 2 * H, 2 * W), device=x.device, dtype=torch.float32)
        if skip is not None:
            skip = skip.contiguous()
        L.check(lib.medt_up2x_relu_add_fwd(x.data_ptr(), L.ptr(skip), y.data_ptr(), N * Cc, H, W, _stream()),
                "medt_up2x_relu_add_fwd")
        ctx.save_for_backward(x)
        ctx.has_skip = skip is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.lib()
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        N, Cc, H, W = x.shape
        dx = torch.empty_like(x)
        L.check(lib.medt_up2x_relu_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), N * Cc, H, W, _stream()),
                "medt_up2x_relu_bwd")
        dskip = dy if ctx.has_skip else None
        if dskip is not None and ctx.skip_sink is not None and ctx.skip_sink.deposit(dskip):
            dskip = None                               # the skip tensor's block adds it in its dgrad epilogue (GradSink)
        return dx, dskip, None


def up2x_relu_add(x, skip=None, skip_sink=None):
    return UpReluAddFn.apply(x, skip, skip_sink)


# --------------------------------------------------------------------------- #
# gate network of AxialAttention_gated_data
# --------------------------------------------------------------------------- #
class GateMlpFn(torch.autograd.Function):
    """(N,C,H,W) -> (B*, 4) per-sequence gates: sigmoid(relu(fcn2(relu(fcn1(mean over the sequence of x)))))
    (reference lib/models/model_codes.py:371-380)."""

    @staticmethod
    def forward(ctx, x, w1, b1, w2, b2, axis):
        _require_device(x)
        lib = L.lib()
        x = x.contiguous()
        N, Cc, H, W = x.shape
        nseq = N * (H if axis else W)
        dev = x.device
        xn = torch.empty((nseq, Cc), device=dev, dtype=torch.float32)
        h = torch.empty_like(xn)
        o = torch.empty((nseq, 4), device=dev, dtype=torch.float32)
        gates = torch.empty_like(o)
        L.check(lib.medt_gate_mlp_fwd(x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                                      xn.data_ptr(), h.data_ptr(), o.data_ptr(), gates.data_ptr(), N, Cc, H, W, axis,
                                      _stream()), "medt_gate_mlp_fwd")
        ctx.save_for_backward(w1, w2, xn, h, o, gates)
        ctx.geom = (N, Cc, H, W, axis)
        return gates

    @staticmethod
    def backward(ctx, dgates):
        lib = L.lib()
        w1, w2, xn, h, o, gates = ctx.saved_tensors
        N, Cc, H, W, axis = ctx.geom
        dev = dgates.device
        nseq = gates.shape[0]
        dgates = dgates.contiguous()
        scratch = torch.empty((nseq * (4 + 2 * Cc),), device=dev, dtype=torch.float32)
        dw1, db1 = torch.empty_like(w1), torch.empty((Cc,), device=dev, dtype=torch.float32)
        dw2, db2 = torch.empty_like(w2), torch.empty((4,), device=dev, dtype=torch.float32)
        dx = torch.empty((N, Cc, H, W), device=dev, dtype=torch.float32)
        L.check(lib.medt_gate_mlp_bwd(dgates.data_ptr(), gates.data_ptr(), o.data_ptr(), h.data_ptr(), xn.data_ptr(),
                                      w1.data_ptr(), w2.data_ptr(), scratch.data_ptr(), dw1.data_ptr(), db1.data_ptr(),
                                      dw2.data_ptr(), db2.data_ptr(), dx.data_ptr(), N, Cc, H, W, axis, _stream()),
                "medt_gate_mlp_bwd")
        return dx, dw1, db1, dw2, db2, None


def gate_mlp(x, fcn1, fcn2, width: bool):
    return GateMlpFn.apply(x, fcn1.weight, fcn1.bias, fcn2.weight, fcn2.bias, 1 if width else 0)


# --------------------------------------------------------------------------- #
# LoGo patches
# --------------------------------------------------------------------------- #
def patch_gather(x, P=32, G=4):
    """(N,C,S,S) image -> (G*G*N, C, P, P) patch-major stack.  Input images carry no gradient."""
    _require_device(x)
    x = x.contiguous()
    N, Cc, S, _ = x.shape
    xp = torch.empty((G * G * N, Cc, P, P), device=x.device, dtype=torch.float32)
    L.check(L.lib().medt_patch_gather(x.data_ptr(), xp.data_ptr(), N, Cc, S, P, G, _stream()), "medt_patch_gather")
    return xp


class LogoMergeFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, yp, P, G):
        _require_device(x)
        x, yp = x.contiguous(), yp.contiguous()
        N, Cc, S, _ = x.shape
        y = torch.empty_like(x)
        L.check(L.lib().medt_logo_merge_fwd(x.data_ptr(), yp.data_ptr(), y.data_ptr(), N, Cc, S, P, G, _stream()),
                "medt_logo_merge_fwd")
        ctx.geom = (N, Cc, S, P, G)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, Cc, S, P, G = ctx.geom
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        dyp = torch.empty((G * G * N, Cc, P, P), device=dy.device, dtype=torch.float32)
        L.check(L.lib().medt_logo_merge_bwd(dy.data_ptr(), dx.data_ptr(), dyp.data_ptr(), N, Cc, S, P, G, _stream()),
                "medt_logo_merge_bwd")
        return dx, dyp, None, None


def logo_merge(x, yp, P=32, G=4):
    return LogoMergeFn.apply(x, yp, P, G)


# --------------------------------------------------------------------------- #
# cross entropy
# --------------------------------------------------------------------------- #
class CrossEntropyFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, ignore_index):
        _require_device(logits)
        lib = L.lib()
        logits, target = logits.contiguous(), target.contiguous()
        if target.dtype != torch.int64:
            raise L.MedtError("cross_entropy: int64 class-index targets expected")
        N, K = logits.shape[0], logits.shape[1]
        HW = logits[0, 0].numel()
        partials = torch.empty((lib.medt_ce_partials(N, HW),), device=logits.device, dtype=torch.float32)
        out = torch.empty((3,), device=logits.device, dtype=torch.float32)
        L.check(lib.medt_ce_fwd(logits.data_ptr(), target.data_ptr(), partials.data_ptr(), out.data_ptr(), N, K, HW,
                                ignore_index, _stream()), "medt_ce_fwd")
        ctx.save_for_backward(logits, target, out)
        ctx.ignore_index = ignore_index
        loss = out[0]
        ctx.mark_non_differentiable(out)
        ctx.set_materialize_grads(False)          # no zero-fill launch for the gradient of the non-differentiable output
        return loss, out

    @staticmethod
    def backward(ctx, dloss, _dout):
        if dloss is None:
            return None, None, None
        lib = L.lib()
        logits, target, out = ctx.saved_tensors
        N, K = logits.shape[0], logits.shape[1]
        HW = logits[0, 0].numel()
        dloss = dloss.contiguous().float()
        dlogits = torch.empty_like(logits)
        L.check(lib.medt_ce_bwd(logits.data_ptr(), target.data_ptr(), out.data_ptr(), dloss.data_ptr(),
                                dlogits.data_ptr(), N, K, HW, ctx.ignore_index, _stream()), "medt_ce_bwd")
        return dlogits, None, None


CHECK_TARGETS = os.environ.get("MEDT_CHECK_TARGETS", "1") != "0"


def cross_entropy(logits, target, ignore_index=-100):
    """F.cross_entropy(logits, target) with mean reduction (what LogNLLLoss.forward computes, metrics.py:17-20).

    Class indices outside [0, K) that are not `ignore_index` make torch raise; here the kernel counts them and this
    wrapper raises MedtError from the count (one host sync; skipped while a hipGraph is being captured -- TrainStep
    checks the same counter after the replay -- and when MEDT_CHECK_TARGETS=0)."""
    loss, out = CrossEntropyFn.apply(logits, target, ignore_index)
    loss._medt_ce_out = out                       # [mean loss, counted pixels, out-of-range targets]
    if CHECK_TARGETS and not torch.cuda.is_current_stream_capturing():
        raise_on_bad_targets(out, logits.shape[1])
    return loss


def raise_on_bad_targets(out, K):
    bad = int(out[2].item())
    if bad:
        rng = f"[0, {K})" if K > 0 else "[0, num_classes)"
        raise L.MedtError(f"cross_entropy: {bad} target value(s) outside {rng} that are not ignore_index "
                          "(torch.nn.functional.cross_entropy raises on these too)")


# --------------------------------------------------------------------------- #
# segmentation scoring (replaces performancemetrics_*.m)
# --------------------------------------------------------------------------- #
def seg_counts(logits, target, threshold=0.5):
    """(N,K,H,W) logits, (N,H,W) int64 labels -> (N,4) int32 {tp, fp, fn, tn} of `logits[:,1] >= threshold` vs
    `target > 0`, counted on the device."""
    _require_device(logits)
    logits, target = logits.contiguous(), target.contiguous()
    if target.dtype != torch.int64:
        raise L.MedtError("seg_counts: int64 label maps expected")
    N, K = logits.shape[0], logits.shape[1]
    HW = logits[0, 0].numel()
    counts = torch.empty((N, 4), device=logits.device, dtype=torch.int32)
    L.check(L.lib().medt_seg_counts(logits.data_ptr(), target.data_ptr(), counts.data_ptr(), N, K, HW, float(threshold),
                                    _stream()), "medt_seg_counts")
    return counts
