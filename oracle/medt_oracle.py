"""CPU oracle for the Medical-Transformer gated axial-attention hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The product path (``medical-transformer_amd/``) never imports this module
and has no CPU fallback.

It is a *functional* restatement, in plain torch on CPU tensors (fp32 or fp64),
of what the reference computes on the path named by BASELINE.json:

  * the three axial-attention layers      (reference lib/models/axialnet.py:19-258)
  * the three residual axial blocks       (reference lib/models/axialnet.py:262-391)
  * the two networks + four factories     (reference lib/models/axialnet.py:397-728)
  * the loss                              (reference metrics.py:17-20)
  * one optimiser step                    (reference train.py:111-112,159-161)

The reference is written as nn.Modules that materialise gathered (2gp,L,L)
embeddings, cat'ed (B*,3G,L,L) logits and nn.BatchNorm calls.  Here the same
arithmetic is written as closed-form tensor algebra over a flat ``state``
dict (tensor names are the reference's state_dict keys): BatchNorm is spelled
out as mean / biased-variance normalisation with the running-stat recurrence,
the relative-position terms index ``relative[c, i-j+L-1]`` directly, and the
networks are walked by key prefix instead of by module.  Parity pinning: the
reference ships no tests or golden vectors for this path (SURVEY.md section
8c), so the oracle is pinned by executing the reference itself
(tests/test_oracle_vs_reference.py, run where /root/reference exists) and by
the fixtures under tests/golden/ that were generated from the reference by
tests/golden/make_golden.py.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

State = Dict[str, torch.Tensor]

BN_EPS = 1e-5        # nn.BatchNorm default, used everywhere in axialnet.py
BN_MOMENTUM = 0.1    # nn.BatchNorm default
GROUPS = 8           # axialnet.py:400,512 (groups=8 for every attention layer)

# bench.py's cpu_baseline leg times this restatement as a stand-in for the reference's CPU path.
# The spelled-out BatchNorm below is ~3x slower on CPU than the fused aten kernel the reference's
# nn.BatchNorm modules hit, which would flatter the GPU/CPU ratio; FAST_BN=True routes batch_norm()
# through torch.nn.functional.batch_norm (same arithmetic, checked equal in tests/test_oracle_golden.py).
FAST_BN = False


def set_fast_bn(flag: bool) -> None:
    global FAST_BN
    FAST_BN = bool(flag)


# --------------------------------------------------------------------------- #
# BatchNorm, spelled out
# --------------------------------------------------------------------------- #
def batch_norm(x: torch.Tensor, st: State, prefix: str, training: bool,
               bn_groups: int = 1) -> torch.Tensor:
    """nn.BatchNorm{1,2}d over channel dim 1 (train: batch stats + running-stat
    update in ``st``; eval: running stats).

    ``bn_groups`` > 1 splits the batch dim 0 into that many consecutive groups,
    each normalised with its own statistics, with the running-stat recurrence
    applied group after group -- this is what the reference's 16-iteration
    patch loop does (axialnet.py:661-700) when the patches
    are stacked patch-major on dim 0.
    """
    w, b = st[prefix + ".weight"], st[prefix + ".bias"]
    C = x.shape[1]
    shape = [1] * x.dim()
    shape[1] = C
    if not training:
        mean, var = st[prefix + ".running_mean"], st[prefix + ".running_var"]
        return (x - mean.view(shape)) * torch.rsqrt(var.view(shape) + BN_EPS) * w.view(shape) + b.view(shape)
    if FAST_BN:
        outs = []
        for xg in x.chunk(bn_groups, dim=0):
            rm, rv = st[prefix + ".running_mean"], st[prefix + ".running_var"]
            outs.append(F.batch_norm(xg, rm, rv, w, b, True, BN_MOMENTUM, BN_EPS))      # updates rm / rv in place
            st[prefix + ".num_batches_tracked"] = st[prefix + ".num_batches_tracked"] + 1
        return torch.cat(outs, 0) if bn_groups > 1 else outs[0]
    outs = []
    for xg in x.chunk(bn_groups, dim=0):
        dims = [d for d in range(xg.dim()) if d != 1]
        n = xg.numel() // C
        if n <= 1:
            raise ValueError("Expected more than 1 value per channel when training")
        mean = xg.mean(dim=dims)
        var = xg.var(dim=dims, unbiased=False)
        with torch.no_grad():
            rm, rv = st[prefix + ".running_mean"], st[prefix + ".running_var"]
            st[prefix + ".running_mean"] = (1 - BN_MOMENTUM) * rm + BN_MOMENTUM * mean.detach().to(rm.dtype)
            st[prefix + ".running_var"] = (1 - BN_MOMENTUM) * rv + BN_MOMENTUM * (var.detach() * n / (n - 1)).to(rv.dtype)
            st[prefix + ".num_batches_tracked"] = st[prefix + ".num_batches_tracked"] + 1
        outs.append((xg - mean.view(shape)) * torch.rsqrt(var.view(shape) + BN_EPS) * w.view(shape) + b.view(shape))
    return torch.cat(outs, 0) if bn_groups > 1 else outs[0]


# --------------------------------------------------------------------------- #
# The attention layer (reference axialnet.py:52-92, 142-189, 222-253)
# --------------------------------------------------------------------------- #
def attention_kind(st: State, prefix: str) -> str:
    if prefix + ".relative" not in st:
        return "wopos"                         # AxialAttention_wopos   (:195)
    if prefix + ".f_qr" in st:
        return "dynamic"                       # AxialAttention_dynamic (:99)
    return "plain"                             # AxialAttention         (:19)


def axial_attention(x: torch.Tensor, st: State, prefix: str, width: bool, stride: int,
                    training: bool, bn_groups: int = 1, taps: Optional[dict] = None,
                    gate_mode: str = "raw") -> torch.Tensor:
    """One axial-attention layer on an NCHW tensor.

    width=False attends along H (one sequence per (n, w)); width=True along W.
    ``taps`` (optional dict) receives the intermediate tensors the kernels
    exchange (qkv after bn_qkv, the logits, the stacked sv|sve) for unit tests.
    ``gate_mode`` selects the experimental gate flavours of the reference's model_codes.py:
    "raw" = axialnet.py (gates multiply as stored), "sigmoid" = AxialAttention_gated_sig (model_codes.py:215-313:
    sigmoid(f) multiplies, :279-280,292-293), "data" = AxialAttention_gated_data (:316-443: four gates per SEQUENCE
    from sigmoid(relu(fcn2(relu(fcn1(mean_L x))))), :371-380,406-407,420-421).
    """
    kind = attention_kind(st, prefix)
    if gate_mode == "data":
        kind = "dynamic"
    N, C, H, W = x.shape
    G = GROUPS
    gp = C // G
    hq = gp // 2
    # (B*, C, L) with B* = N*W (height layer) or N*H (width layer)  -- :143-148
    X = x.permute(0, 2, 1, 3) if width else x.permute(0, 3, 1, 2)
    Bo = X.shape[1]
    L = X.shape[3]
    X = X.reshape(N * Bo, C, L)
    B = N * Bo
    if gate_mode == "sigmoid":
        f_qr, f_kr, f_sv, f_sve = (torch.sigmoid(st[prefix + n]) for n in (".f_qr", ".f_kr", ".f_sv", ".f_sve"))
    elif gate_mode == "data":
        xn = X.mean(dim=2)                                                     # AdaptiveAvgPool2d((1,1)) over L (:371)
        xn = torch.relu(F.linear(xn, st[prefix + ".fcn1.weight"], st[prefix + ".fcn1.bias"]))
        xn = torch.relu(F.linear(xn, st[prefix + ".fcn2.weight"], st[prefix + ".fcn2.bias"]))
        sig = torch.sigmoid(xn).reshape(B, 4, 1, 1, 1)
        f_qr, f_kr, f_sv, f_sve = sig[:, 0], sig[:, 1], sig[:, 2], sig[:, 3]   # sig3 -> sv, sig4 -> sve (:420-421)
    elif kind == "dynamic":
        f_qr, f_kr, f_sv, f_sve = (st[prefix + n] for n in (".f_qr", ".f_kr", ".f_sv", ".f_sve"))
    # BN groups follow the image index n, which is the slow part of b = n*Bo + s
    Wqkv = st[prefix + ".qkv_transform.weight"].reshape(2 * C, C)          # Conv1d k=1, no bias (:114)
    qkv = torch.einsum("oc,bcl->bol", Wqkv, X)
    qkv = batch_norm(qkv, st, prefix + ".bn_qkv", training, bn_groups)        # :151
    qkv4 = qkv.reshape(B, G, 2 * gp, L)
    q, k, v = qkv4[:, :, :hq], qkv4[:, :, hq:gp], qkv4[:, :, gp:]
    qk = torch.einsum("bgci,bgcj->bgij", q, k)                                  # :159
    if kind == "wopos":
        Z = batch_norm(qk, st, prefix + ".bn_similarity", training, bn_groups)  # BN2d(G)  :236
    else:
        R = st[prefix + ".relative"]                                            # (2gp, 2L-1)
        if R.shape[1] != 2 * L - 1:
            raise RuntimeError(f"relative table built for L={(R.shape[1] + 1) // 2}, sequence has L={L}")
        ar = torch.arange(L)
        d = ar.view(L, 1) - ar.view(1, L) + (L - 1)                             # d[i,j] = i-j+L-1 (:132-135)
        Rq, Rk, Rv = R[:hq], R[hq:gp], R[gp:]
        qr = torch.einsum("bgci,cij->bgij", q, Rq[:, d])                        # q[c,i]*Rq[c,i-j+L-1]
        kr = torch.einsum("bgcj,cij->bgij", k, Rk[:, d.t()])                    # k[c,j]*Rk[c,j-i+L-1]  (:158)
        if kind == "dynamic":
            qr = qr * f_qr                                                      # :163-164
            kr = kr * f_kr
        S = torch.cat([qk, qr, kr], dim=1)                                      # channel order qk|qr|kr (:166)
        S = batch_norm(S, st, prefix + ".bn_similarity", training, bn_groups)   # BN2d(3G)
        Z = S[:, :G] + S[:, G:2 * G] + S[:, 2 * G:]                             # .view(B,3,G,L,L).sum(1)
    P = torch.softmax(Z, dim=3)                                                 # over keys j  (:170)
    sv = torch.einsum("bgij,bgcj->bgci", P, v)                                  # :171
    if kind == "wopos":
        stacked = sv.reshape(B, C, L)                                           # :241
        out = batch_norm(stacked, st, prefix + ".bn_output", training, bn_groups)  # BN1d(C)
    else:
        sve = torch.einsum("bgij,cij->bgci", P, Rv[:, d])                       # :172
        if kind == "dynamic":
            sv = sv * f_sv                                                      # :175-176
            sve = sve * f_sve
        # cat(dim=-1).view(B, 2C, L): channel 2*(g*gp+c)+0 <- sv, +1 <- sve   (:178)
        stacked = torch.stack([sv, sve], dim=3).reshape(B, 2 * C, L)
        out = batch_norm(stacked, st, prefix + ".bn_output", training, bn_groups)  # BN1d(2C)
        out = out.reshape(B, C, 2, L).sum(dim=2)                                # pair-sum (:179)
    if taps is not None:
        taps.update(qkv=qkv, Z=Z, P=P, stacked=stacked)
    out = out.reshape(N, Bo, C, L)
    out = out.permute(0, 2, 1, 3) if width else out.permute(0, 2, 3, 1)        # :181-184
    if stride > 1:
        out = F.avg_pool2d(out, stride, stride)                                 # :186-187
    return out


# --------------------------------------------------------------------------- #
# Blocks and networks
# --------------------------------------------------------------------------- #
def _conv(x, st, prefix, stride=1, padding=0):
    return F.conv2d(x, st[prefix + ".weight"], st.get(prefix + ".bias"), stride=stride, padding=padding)


def axial_block(x, st: State, prefix: str, stride: int, training: bool, bn_groups: int = 1):
    """AxialBlock / _dynamic / _wopos forward (axialnet.py:282-302, 324-344, 368-391)."""
    out = _conv(x, st, prefix + ".conv_down")
    out = torch.relu(batch_norm(out, st, prefix + ".bn1", training, bn_groups))
    out = axial_attention(out, st, prefix + ".hight_block", False, 1, training, bn_groups)
    out = axial_attention(out, st, prefix + ".width_block", True, stride, training, bn_groups)
    out = torch.relu(out)
    out = batch_norm(_conv(out, st, prefix + ".conv_up"), st, prefix + ".bn2", training, bn_groups)
    if prefix + ".downsample.0.weight" in st:
        identity = _conv(x, st, prefix + ".downsample.0", stride=stride)
        identity = batch_norm(identity, st, prefix + ".downsample.1", training, bn_groups)
    else:
        identity = x
    return torch.relu(out + identity)


def _layer(x, st, prefix, nblocks, stride, training, bn_groups=1):
    for b in range(nblocks):
        x = axial_block(x, st, f"{prefix}.{b}", stride if b == 0 else 1, training, bn_groups)
    return x


def _up(x):
    # F.interpolate(scale_factor=(2,2), mode='bilinear'), align_corners=False default (Q10)
    return F.interpolate(x, scale_factor=(2, 2), mode="bilinear", align_corners=False)


LAYERS = (1, 2, 4, 1)            # all four factories (axialnet.py:714-728)


def _stem(x, st, sfx, training, bn_groups=1):
    x = torch.relu(batch_norm(_conv(x, st, "conv1" + sfx, stride=2, padding=3), st, "bn1" + sfx, training, bn_groups))
    x = torch.relu(batch_norm(_conv(x, st, "conv2" + sfx, padding=1), st, "bn2" + sfx, training, bn_groups))
    x = torch.relu(batch_norm(_conv(x, st, "conv3" + sfx, padding=1), st, "bn3" + sfx, training, bn_groups))
    return x


def _unet_body(x, st, sfx, training, bn_groups=1):
    """layer1..4 + decoder1..5 with skips (axialnet.py:485-501 and 682-698)."""
    x1 = _layer(x, st, "layer1" + sfx, LAYERS[0], 1, training, bn_groups)
    x2 = _layer(x1, st, "layer2" + sfx, LAYERS[1], 2, training, bn_groups)
    x3 = _layer(x2, st, "layer3" + sfx, LAYERS[2], 2, training, bn_groups)
    x4 = _layer(x3, st, "layer4" + sfx, LAYERS[3], 2, training, bn_groups)
    y = torch.relu(_up(_conv(x4, st, "decoder1" + sfx, stride=2, padding=1))) + x4
    y = torch.relu(_up(_conv(y, st, "decoder2" + sfx, padding=1))) + x3
    y = torch.relu(_up(_conv(y, st, "decoder3" + sfx, padding=1))) + x2
    y = torch.relu(_up(_conv(y, st, "decoder4" + sfx, padding=1))) + x1
    y = torch.relu(_up(_conv(y, st, "decoder5" + sfx, padding=1)))
    return y


def res_axial_unet(x, st: State, training: bool):
    """ResAxialAttentionUNet._forward_impl (axialnet.py:471-504): `gated`, `axialunet`."""
    y = _unet_body(_stem(x, st, "", training), st, "", training)
    return _conv(torch.relu(y), st, "adjust")


def medt(x, st: State, training: bool, batch_patches: bool = False):
    """medt_net._forward_impl (axialnet.py:620-708): `MedT`, `logo`.

    ``batch_patches=True`` evaluates the 16 patches stacked patch-major on the
    batch dim with 16 BN groups -- mathematically the reference's sequential
    loop (the product's batched layout); False walks the loop literally.
    """
    xin = x
    g = _stem(x, st, "", training)
    x1 = _layer(g, st, "layer1", LAYERS[0], 1, training)
    x2 = _layer(x1, st, "layer2", LAYERS[1], 2, training)
    y = torch.relu(_up(_conv(x2, st, "decoder4", padding=1))) + x1
    y = torch.relu(_up(_conv(y, st, "decoder5", padding=1)))
    x_loc = y.clone()
    N = x.shape[0]
    if batch_patches:
        patches = [xin[:, :, 32 * i:32 * i + 32, 32 * j:32 * j + 32] for i in range(4) for j in range(4)]
        xp = torch.cat(patches, 0)                                   # (16N, c, 32, 32), patch-major
        yp = _unet_body(_stem(xp, st, "_p", training, 16), st, "_p", training, 16)
        for p in range(16):
            i, j = divmod(p, 4)
            x_loc[:, :, 32 * i:32 * i + 32, 32 * j:32 * j + 32] = yp[p * N:(p + 1) * N]
    else:
        for i in range(4):                                           # hard-coded 4x4 grid of 32-px patches (:661-664)
            for j in range(4):
                xp = xin[:, :, 32 * i:32 * i + 32, 32 * j:32 * j + 32]
                yp = _unet_body(_stem(xp, st, "_p", training), st, "_p", training)
                x_loc[:, :, 32 * i:32 * i + 32, 32 * j:32 * j + 32] = yp
    y = y + x_loc
    y = torch.relu(_conv(y, st, "decoderf", padding=1))
    return _conv(torch.relu(y), st, "adjust")


def forward(model_name: str, x, st: State, training: bool, **kw):
    """model_name in the CLI vocabulary of train.py:95-102."""
    if model_name in ("gatedaxialunet", "gated", "axialunet"):
        return res_axial_unet(x, st, training)
    if model_name in ("MedT", "logo"):
        return medt(x, st, training, **kw)
    raise ValueError(model_name)


# --------------------------------------------------------------------------- #
# Loss and optimiser step
# --------------------------------------------------------------------------- #
def log_nll_loss(logits, target):
    """LogNLLLoss.forward = plain mean cross entropy (metrics.py:17-20)."""
    lse = torch.logsumexp(logits, dim=1)
    picked = logits.gather(1, target.unsqueeze(1)).squeeze(1)
    return (lse - picked).mean()


def adam_step(p, g, m, v, step, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8, wd=1e-5):
    """torch.optim.Adam with coupled L2 weight decay (train.py:111-112). Returns new (p, m, v)."""
    g = g + wd * p
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    return p - (lr / bc1) * m / denom, m, v


# --------------------------------------------------------------------------- #
# helpers for tests
# --------------------------------------------------------------------------- #
def clone_state(sd: State, dtype=None, requires_grad: bool = False) -> State:
    out = {}
    for k, t in sd.items():
        t = t.detach().clone()
        if t.is_floating_point():
            if dtype is not None:
                t = t.to(dtype)
            if requires_grad and not (k.endswith("running_mean") or k.endswith("running_var")):
                t.requires_grad_(True)
        out[k] = t
    return out


def randomize_state(sd: State, seed: int) -> State:
    """Deterministic non-trivial values for every entry of a state_dict, by key order.

    Default inits leave BN weights at 1, biases at 0, running stats at 0/1 and
    the gates at 0.1/1.0, which would hide indexing mistakes; this fills every
    float tensor from a seeded CPU generator so fixtures generated in one
    container can be regenerated bit-identically in another.
    """
    g = torch.Generator().manual_seed(seed)
    out = {}
    for k, t in sd.items():
        if not t.is_floating_point():
            out[k] = t.clone()
            continue
        r = torch.randn(t.shape, generator=g, dtype=torch.float32)
        if k.endswith("running_var"):
            val = 0.5 + r.abs()
        elif k.endswith("running_mean"):
            val = 0.2 * r
        elif ".bn" in k or k.startswith("bn") or ".downsample.1" in k:
            val = (1.0 + 0.2 * r) if k.endswith("weight") else 0.1 * r
        elif k.endswith((".f_qr", ".f_kr", ".f_sve", ".f_sv")):
            val = 0.5 + 0.25 * r
        elif k.endswith(".relative"):
            val = r * math.sqrt(1.0 / (t.shape[0] / 2))
        elif k.endswith("qkv_transform.weight"):
            val = r * math.sqrt(1.0 / t.shape[1])
        elif t.dim() >= 2:
            fan_in = t[0].numel()
            val = r * math.sqrt(1.0 / fan_in)
        else:
            val = 0.05 * r
        out[k] = val.to(t.dtype)
    return out
