"""Host side of the axial-attention layer: torch.autograd.Function over the C ABI.

Mirrors what AxialAttention{,_dynamic,_wopos}.forward computes
(reference lib/models/axialnet.py:52-92, 142-189, 222-253) and its autograd
backward; PyTorch only supplies device memory, the stream and the autograd
graph.  CPU tensors are rejected -- there is no fallback path.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib as L
from . import optim as OPT
from . import defer as DEFER


ACT_BF16 = False


def set_activation_dtype(dtype) -> None:
    """Storage type of the attention layers' saved activations (qkv_transform output and sv|sve): torch.float32 -- the
    reference's -- or torch.bfloat16 (BASELINE.json configs[1]).  Arithmetic, statistics, layer inputs/outputs and all
    gradients stay float32 either way; position-free (wopos) layers always store float32."""
    global ACT_BF16
    if dtype in (torch.bfloat16, "bf16", "bfloat16"):
        ACT_BF16 = True
    elif dtype in (torch.float32, "f32", "fp32", "float32"):
        ACT_BF16 = False
    else:
        raise L.MedtError(f"activation storage dtype {dtype!r} unsupported (float32 or bfloat16)")


class AxialConfig:
    """Static geometry + BatchNorm buffers of one attention layer."""
    __slots__ = ("groups", "axis", "has_pos", "stride", "bn_groups", "eps", "momentum",
                 "bn_qkv", "bn_similarity", "bn_output", "out_relu", "gate_mode", "act_dtype", "pre")

    def __init__(self, groups, axis, has_pos, stride, bn_qkv, bn_similarity, bn_output, bn_groups=1,
                 eps=1e-5, momentum=0.1, out_relu=False, gate_mode=0):
        self.groups, self.axis, self.has_pos, self.stride = groups, axis, has_pos, stride
        self.gate_mode = gate_mode
        # storage of the layer's saved qkv_raw / stacked tensors: bf16 only for the position-encoded layers
        self.act_dtype = 1 if (ACT_BF16 and has_pos) else 0
        self.bn_groups, self.eps, self.momentum, self.out_relu = bn_groups, eps, momentum, out_relu
        self.bn_qkv, self.bn_similarity, self.bn_output = bn_qkv, bn_similarity, bn_output
        self.pre = None       # (qkv_raw, stacked, lse, stats, y) already computed by the one-launch block forward (block.py)


def _require_device(x: torch.Tensor):
    if not x.is_cuda:
        raise L.MedtError("medt_amd runs on MI355X only: got a CPU tensor and there is no CPU fallback "
                          "(the CPU restatement under oracle/ is test infrastructure)")
    if x.dtype != torch.float32:
        raise L.MedtError(f"medt_amd: float32 activations expected, got {x.dtype}")


def _momentum(bn) -> float:
    if bn.momentum is None:
        raise L.MedtError("BatchNorm momentum=None (cumulative moving average) is not implemented by the HIP kernels; "
                          "the reference uses the default 0.1 everywhere")
    return float(bn.momentum)


def _bn_ptrs(bn, training: bool) -> L.BnPtrs:
    track = bn.running_mean is not None
    return L.BnPtrs(L.ptr(bn.weight), L.ptr(bn.bias),
                    L.ptr(bn.running_mean) if track else None, L.ptr(bn.running_var) if track else None,
                    L.ptr(bn.num_batches_tracked) if (track and training) else None)


def _desc(x, cfg: AxialConfig, training: bool) -> L.AxialDesc:
    N, Cc, H, W = x.shape
    return L.AxialDesc(N, Cc, H, W, cfg.groups, cfg.axis, int(cfg.has_pos), cfg.stride, int(training),
                       cfg.bn_groups, cfg.eps, cfg.momentum, int(cfg.out_relu), int(cfg.gate_mode), int(cfg.act_dtype))


def _params(cfg, w_qkv, relative, gates, training) -> L.AxialParams:
    g = [L.ptr(t) for t in gates] if gates is not None else [None] * 4
    return L.AxialParams(L.ptr(w_qkv), _bn_ptrs(cfg.bn_qkv, training), _bn_ptrs(cfg.bn_similarity, training),
                         _bn_ptrs(cfg.bn_output, training), L.ptr(relative), g[0], g[1], g[2], g[3])


class AxialAttentionFn(torch.autograd.Function):
    """y = axial_attention(x).  Argument order = gradient order."""

    @staticmethod
    def forward(ctx, x, w_qkv, bnq_w, bnq_b, bns_w, bns_b, bno_w, bno_b, relative, f_qr, f_kr, f_sve, f_sv,
                cfg: AxialConfig, training: bool):
        _require_device(x)
        lib = L.lib()
        x = x.contiguous()
        N, Cc, H, W = x.shape
        desc = _desc(x, cfg, training)
        gates = None if f_qr is None else (f_qr, f_kr, f_sve, f_sv)
        params = _params(cfg, w_qkv, relative, gates, training)
        OC = 2 * Cc if cfg.has_pos else Cc
        dev = x.device
        sdt = torch.bfloat16 if cfg.act_dtype == 1 else torch.float32
        if cfg.pre is not None:                    # adopt mode: the one-launch block forward already produced these
            qkv_raw, stacked, lse, stats, y = cfg.pre
            cfg.pre = None
        else:
            qkv_raw = torch.empty((N, 2 * Cc, H, W), device=dev, dtype=sdt)
            stacked = torch.empty((N, OC, H, W), device=dev, dtype=sdt)
            lse = torch.empty((N, cfg.groups, H, W), device=dev, dtype=torch.float32)
            nstats = lib.medt_axial_stats_floats(C.byref(desc))
            if nstats == 0:
                raise L.MedtError("axial attention: " + lib.medt_last_error().decode())
            stats = torch.empty((nstats,), device=dev, dtype=torch.float32)
            ws_bytes = lib.medt_axial_workspace_bytes(C.byref(desc))
            ws = torch.empty((ws_bytes,), device=dev, dtype=torch.uint8)
            y = torch.empty((N, Cc, H // cfg.stride, W // cfg.stride), device=dev, dtype=torch.float32)
            saved = L.AxialSaved(qkv_raw.data_ptr(), stacked.data_ptr(), lse.data_ptr(), stats.data_ptr())
            stream = torch.cuda.current_stream().cuda_stream
            q = DEFER.recording()
            L.check(lib.medt_axial_layer_fwd(C.byref(desc), C.byref(params), x.data_ptr(), y.data_ptr(), C.byref(saved),
                                             ws.data_ptr(), ws_bytes, stream), "medt_axial_layer_fwd")
            if q is not None:
                q.hold(ws, stats)
        ctx.cfg, ctx.training, ctx.has_gates = cfg, training, gates is not None
        # gradient slots of the parameters (views into FlatAdam's flat bucket): backward writes them directly
        ctx.slots = tuple(OPT.grad_slot(t) if t is not None else None
                          for t in (w_qkv, bnq_w, bnq_b, bns_w, bns_b, bno_w, bno_b, relative))
        ctx.gate_slots = tuple(OPT.grad_slot(t) for t in gates) if gates is not None else None
        ctx.save_for_backward(x, w_qkv, relative, f_qr, f_kr, f_sve, f_sv, qkv_raw, stacked, lse, stats,
                              y if cfg.out_relu else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = L.lib()
        x, w_qkv, relative, f_qr, f_kr, f_sve, f_sv, qkv_raw, stacked, lse, stats, y = ctx.saved_tensors
        cfg, training = ctx.cfg, ctx.training
        dy = dy.contiguous()
        desc = _desc(x, cfg, training)
        gates = (f_qr, f_kr, f_sve, f_sv) if ctx.has_gates else None
        params = _params(cfg, w_qkv, relative, gates, False)      # no running-stat update in backward
        dev = x.device
        Cc = x.shape[1]
        SC = (3 if cfg.has_pos else 1) * cfg.groups
        OC = (2 if cfg.has_pos else 1) * Cc
        sizes = [2 * Cc * Cc, 2 * Cc, 2 * Cc, SC, SC, OC, OC, relative.numel() if relative is not None else 0]
        shapes = [w_qkv.shape, None, None, None, None, None, None, relative.shape if relative is not None else None]
        # destinations: the parameter's gradient slot in FlatAdam's flat bucket when it has one (nothing is returned to
        # autograd then), otherwise a slice of one temporary that is handed back as the gradient
        dst, ret, pend, need_tmp = [None] * 8, [None] * 8, [], []
        for k in range(8):
            if sizes[k] == 0:
                continue
            slot = OPT.live(ctx.slots[k])
            if slot is not None and ctx.needs_input_grad[k + 1]:
                dst[k], direct = OPT.claim(slot)
                if not direct:
                    pend.append((slot, dst[k]))
            else:
                need_tmp.append(k)
        want_gates = ctx.has_gates and any(ctx.needs_input_grad[9:13])
        gate_direct = False
        n_gate = f_qr.numel() if (ctx.has_gates and cfg.gate_mode == 2) else 4     # per-sequence gates: a (B*, 4) tensor
        if want_gates and cfg.gate_mode != 2:
            gs = tuple(OPT.live(g_) for g_ in ctx.gate_slots)
            if all(g_ is not None for g_ in gs) and all(ctx.needs_input_grad[9:13]) and \
                    all(gs[i + 1].view.data_ptr() == gs[i].view.data_ptr() + 4 for i in range(3)) and \
                    all(g_.stamp != g_.owner.stamp for g_ in gs):
                for g_ in gs:                                   # four adjacent 0-d slots == the ABI's float[4]
                    OPT.claim(g_)
                gate_direct = True
        tmp_sizes = [sizes[k] for k in need_tmp] + ([n_gate] if (want_gates and not gate_direct) else [])
        if tmp_sizes:
            parts = list(torch.split(torch.empty((sum(tmp_sizes),), device=dev, dtype=torch.float32), tmp_sizes))
            for k, part in zip(need_tmp, parts):
                dst[k] = part
                if ctx.needs_input_grad[k + 1]:
                    ret[k] = part.view(shapes[k]) if shapes[k] is not None else part
        if want_gates:
            gate_ptr = gs[0].view.data_ptr() if gate_direct else parts[-1].data_ptr()
        else:
            gate_ptr = None
        dx = torch.empty_like(x)
        grads = L.AxialGrads(*[L.ptr(t) for t in dst], gate_ptr)
        saved = L.AxialSaved(qkv_raw.data_ptr(), stacked.data_ptr(), lse.data_ptr(), stats.data_ptr())
        ws_bytes = lib.medt_axial_workspace_bytes(C.byref(desc))
        ws = torch.empty((ws_bytes,), device=dev, dtype=torch.uint8)
        stream = torch.cuda.current_stream().cuda_stream
        # parameter gradients may only be recorded for the grouped flush when they all land in persistent slots
        q = DEFER.recording(allow=not pend and all(r is None for r in ret) and (not want_gates or gate_direct))
        L.check(lib.medt_axial_layer_bwd(C.byref(desc), C.byref(params), x.data_ptr(), L.ptr(y), dy.data_ptr(), C.byref(saved),
                                         dx.data_ptr(), C.byref(grads), ws.data_ptr(), ws_bytes, stream),
                "medt_axial_layer_bwd")
        if q is not None:                      # recorded weight-gradient / reduction jobs read these at the flush
            q.hold(ws, x, qkv_raw, stats, *dst)
            if tmp_sizes:
                q.hold(*parts)
        for slot, tmp in pend:
            OPT.accumulate(slot, tmp)
        if want_gates and cfg.gate_mode == 2:
            dg = [parts[-1].view_as(f_qr), None, None, None]
        elif want_gates and not gate_direct:
            gg = parts[-1]
            dg = [gg[i].reshape(()) if ctx.needs_input_grad[9 + i] else None for i in range(4)]
        else:
            dg = [None] * 4
        return (dx, ret[0], ret[1], ret[2], ret[3], ret[4], ret[5], ret[6], ret[7], dg[0], dg[1], dg[2], dg[3],
                None, None)


def axial_attention(x, qkv_weight, bn_qkv, bn_similarity, bn_output, relative: Optional[torch.Tensor],
                    gates, groups: int, width: bool, stride: int, training: bool, bn_groups: int = 1,
                    out_relu: bool = False, gate_mode: int = 0, pre=None):
    """Functional entry: modules from lib.models.axialnet pass their own parameters/buffers.

    gates = (f_qr, f_kr, f_sve, f_sv) 0-d tensors or None (ungated: all ones).
    gate_mode 1: sigmoid(f) multiplies (AxialAttention_gated_sig, reference lib/models/model_codes.py:215-313).
    gate_mode 2: gates = (G, None, None, None) with G a (B*, 4) tensor of per-sequence gates, columns (qr, kr, sv, sve)
    (AxialAttention_gated_data, :316-443; G comes from medt_amd.ops.gate_mlp).
    """
    cfg = AxialConfig(groups, 1 if width else 0, relative is not None, stride, bn_qkv, bn_similarity, bn_output,
                      bn_groups, bn_qkv.eps, _momentum(bn_qkv), out_relu, gate_mode)
    cfg.pre = pre
    g = gates if gates is not None else (None, None, None, None)
    return AxialAttentionFn.apply(x, qkv_weight, bn_qkv.weight, bn_qkv.bias, bn_similarity.weight,
                                  bn_similarity.bias, bn_output.weight, bn_output.bias, relative,
                                  g[0], g[1], g[2], g[3], cfg, training)
