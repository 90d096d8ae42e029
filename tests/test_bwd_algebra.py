"""The algebra behind the single-sweep attention backward (csrc/axial_bwd.hip), checked on the CPU in float64 against
autograd of the reference formulation (reference lib/models/axialnet.py:155-178).

The reference's autograd needs TWO L x L passes in training mode: the bn_similarity backward means
m1 = mean(dY), m2 = mean(dY * xhat) over (B*, L, L) sit between the softmax backward and the gradients of q, k and the
relative table.  The kernels instead run ONE sweep that produces everything that is linear in dZ, and the two means
come out of that sweep for free:

    dS_x = e_x dZ + u_x S_x + w_x          (x in qk, qr, kr;  e_x = gamma_x rstd_x is known before the sweep)
    A_qk = sum dZ S_qk = sum_i q_i . (sum_j dZ_ij k_j)        = q . Dqk      -- the sweep's own dq accumulator
    A_qr = f_qr sum_i q_i . (sum_j dZ_ij Rq[i-j])             = f_qr q . Dqr
    A_kr = f_kr sum_j k_j . (sum_i dZ_ij Rk[j-i])             = f_kr k . Dkr
    sum dZ = 0                                                 (softmax rows)

and the u / w terms do not involve dZ at all: they are closed forms in per-sequence Gram matrices of q and k and in
sliding-window sums of the relative table (the same objects the forward statistics use, csrc/axial_stats.hip):

    sum_j S_qk[i,j] k_c[j]            = sum_c' q_c'[i] Gk[c',c]                    Gk = k k^T per sequence
    sum_j f rq[i,j] Rq_c[i-j+L-1]     = f sum_c' q_c'[i] TQ[c',c][i]               TQ[i] = sum_{d=i}^{i+L-1} Rq_c' Rq_c
    table gradient, u term            = f^2 u sum_c' R_c'[d] sum_{i in win(d)} sum_b q_c'[b,i] q_c[b,i]

This file is test infrastructure: it restates what axial_bwd.hip (sweep), attn_bwd_fix_kernel and attn_bwd_relfix_kernel
compute, in torch, and holds it to autograd at 1e-10."""
import pytest
import torch

EPS = 1e-5


def core_forward(qkv, R, gates, gamma, beta, training=True, run=None):
    """qkv (B, G, 2gp, L) post bn_qkv -> stacked (B, G, gp, 2, L) [sv | sve gated].  gates = (f_qr, f_kr, f_sv, f_sve)."""
    B, G, NCH, L = qkv.shape
    gp = NCH // 2
    hq = gp // 2
    q, k, v = qkv[:, :, :hq], qkv[:, :, hq:gp], qkv[:, :, gp:]
    ar = torch.arange(L)
    d = ar.view(L, 1) - ar.view(1, L) + (L - 1)
    Rq, Rk, Rv = R[:hq], R[hq:gp], R[gp:]
    f_qr, f_kr, f_sv, f_sve = gates
    qk = torch.einsum("bgci,bgcj->bgij", q, k)
    qr = torch.einsum("bgci,cij->bgij", q, Rq[:, d]) * f_qr
    kr = torch.einsum("bgcj,cij->bgij", k, Rk[:, d.t()]) * f_kr
    S = torch.cat([qk, qr, kr], dim=1)                       # (B, 3G, L, L)
    if training:
        mean = S.mean(dim=(0, 2, 3))
        var = S.var(dim=(0, 2, 3), unbiased=False)
    else:
        mean, var = run
    rstd = torch.rsqrt(var + EPS)
    Sn = (S - mean.view(1, -1, 1, 1)) * (rstd * gamma).view(1, -1, 1, 1) + beta.view(1, -1, 1, 1)
    Z = Sn[:, :G] + Sn[:, G:2 * G] + Sn[:, 2 * G:]
    P = torch.softmax(Z, dim=3)
    sv = torch.einsum("bgij,bgcj->bgci", P, v) * f_sv
    sve = torch.einsum("bgij,cij->bgci", P, Rv[:, d]) * f_sve
    return torch.stack([sv, sve], dim=3), dict(P=P, mean=mean.detach(), rstd=rstd.detach())


def single_sweep_backward(qkv, R, gates, gamma, stats, dstk, stk, training=True):
    """What the kernels compute.  Returns dqkv, dR, dgates, dgamma."""
    B, G, NCH, L = qkv.shape
    gp = NCH // 2
    hq = gp // 2
    q, k, v = qkv[:, :, :hq], qkv[:, :, hq:gp], qkv[:, :, gp:]
    f_qr, f_kr, f_sv, f_sve = gates
    Rq, Rk, Rv = R[:hq], R[hq:gp], R[gp:]
    ar = torch.arange(L)
    d = ar.view(L, 1) - ar.view(1, L) + (L - 1)               # d[i, j]
    P, mean, rstd = stats["P"], stats["mean"], stats["rstd"]
    e = (gamma * rstd).view(3, G)                             # e_x per head
    mean3, rstd3 = mean.view(3, G), rstd.view(3, G)
    dsv, dse = dstk[:, :, :, 0], dstk[:, :, :, 1]             # (B, G, gp, L) wrt the gated stacked values
    # ---- the sweep: everything linear in dZ ------------------------------------------------------------------
    delta = (dstk * stk).sum(dim=(2, 3))                      # (B, G, L): sum_j P dP
    dPv = torch.einsum("bgci,bgcj->bgij", dsv, v)
    dPe = torch.einsum("bgci,cij->bgij", dse, Rv[:, d])
    dZ = P * (f_sv * dPv + f_sve * dPe - delta.unsqueeze(-1))
    Dqk = torch.einsum("bgij,bgcj->bgci", dZ, k)              # row accumulators
    Dqr = torch.einsum("bgij,cij->bgci", dZ, Rq[:, d])
    Eqk = torch.einsum("bgij,bgci->bgcj", dZ, q)              # column accumulators
    Dkr = torch.einsum("bgij,cij->bgcj", dZ, Rk[:, d.t()])
    dv = f_sv * torch.einsum("bgij,bgci->bgcj", P, dsv)
    # diagonal accumulators, per head (the per-head scale is applied before the heads are summed)
    onehot = torch.nn.functional.one_hot(d, 2 * L - 1).to(qkv.dtype)          # (L, L, TL)
    aq = torch.einsum("bgij,bgci,ijd->gcd", dZ, q, onehot)    # sum dZ q_c[i]  on diagonal d = i-j+L-1
    ak = torch.einsum("bgij,bgcj,ijd->gcd", dZ, k, onehot.transpose(0, 1))    # sum dZ k_c[j] on d' = j-i+L-1
    av = f_sve * torch.einsum("bgij,bgci,ijd->cd", P, dse, onehot)
    # the bn_similarity backward sums, from the accumulators (no second pass)
    A_qk = (q * Dqk).sum(dim=(0, 2, 3))                       # per head
    A_qr = f_qr * (q * Dqr).sum(dim=(0, 2, 3))
    A_kr = f_kr * (k * Dkr).sum(dim=(0, 2, 3))
    A = torch.stack([A_qk, A_qr, A_kr])                       # (3, G): sum dZ * S_x
    count = B * L * L
    dgamma = (rstd3 * (A - mean3 * 0.0)).reshape(-1)          # sum dZ xhat = rstd (A - mean * sum dZ), sum dZ = 0
    if training:
        m2 = rstd3 * A / count                                # mean(dZ xhat) (per unit gamma)
        u = -e * rstd3 * m2
        w = -u * mean3                                        # m1 = 0
    else:
        u = torch.zeros_like(e)
        w = torch.zeros_like(e)
    u_qk, u_qr, u_kr = u.unbind(0)
    w_qk, w_qr, w_kr = w.unbind(0)
    e_qk, e_qr, e_kr = e.unbind(0)
    hv = lambda t: t.view(1, G, 1, 1)                         # per-head scalar against (B, G, c, L)
    # ---- the fix: closed forms for the u / w terms --------------------------------------------------------------
    Gk = torch.einsum("bgcj,bgej->bgce", k, k)
    Gq = torch.einsum("bgci,bgei->bgce", q, q)
    Sk, Sq = k.sum(-1), q.sum(-1)                             # (B, G, hq)
    win = lambda T: torch.stack([T[..., i:i + L].sum(-1) for i in range(L)], dim=-1)     # sliding-window sums over d
    TQ = win(Rq.unsqueeze(1) * Rq.unsqueeze(0))               # (hq, hq, L): [c', c][i]
    TK = win(Rk.unsqueeze(1) * Rk.unsqueeze(0))
    UQ, UK = win(Rq), win(Rk)                                 # (hq, L)
    dq = (hv(e_qk) * Dqk + hv(f_qr * e_qr) * Dqr
          + hv(u_qk) * torch.einsum("bgei,bgec->bgci", q, Gk) + hv(w_qk) * Sk.unsqueeze(-1)
          + hv(f_qr * f_qr * u_qr) * torch.einsum("bgei,eci->bgci", q, TQ) + hv(f_qr * w_qr) * UQ.view(1, 1, hq, L))
    dk = (hv(e_qk) * Eqk + hv(f_kr * e_kr) * Dkr
          + hv(u_qk) * torch.einsum("bgej,bgec->bgcj", k, Gq) + hv(w_qk) * Sq.unsqueeze(-1)
          + hv(f_kr * f_kr * u_kr) * torch.einsum("bgej,ecj->bgcj", k, TK) + hv(f_kr * w_kr) * UK.view(1, 1, hq, L))
    dqkv = torch.cat([dq, dk, dv], dim=2)
    # ---- the table fix: per-position Gram sums over the sequences, windowed -----------------------------------
    PGq = torch.einsum("bgei,bgci->geci", q, q)               # (G, c', c, L)
    PGk = torch.einsum("bgej,bgcj->gecj", k, k)
    PSq, PSk = q.sum(0), k.sum(0)                             # (G, hq, L)
    TL = 2 * L - 1
    dRq = torch.zeros(hq, TL, dtype=qkv.dtype)
    dRk = torch.zeros(hq, TL, dtype=qkv.dtype)
    for dd in range(TL):
        lo, hi = max(0, dd - L + 1), min(L - 1, dd)           # positions whose diagonal dd lies inside the L x L square
        gq = PGq[..., lo:hi + 1].sum(-1)                      # (G, c', c)
        gk = PGk[..., lo:hi + 1].sum(-1)
        sq, sk = PSq[..., lo:hi + 1].sum(-1), PSk[..., lo:hi + 1].sum(-1)
        dRq[:, dd] = (f_qr * e_qr.view(G, 1) * aq[:, :, dd]).sum(0) \
            + (f_qr * f_qr * u_qr.view(G, 1) * torch.einsum("e,gec->gc", Rq[:, dd], gq)).sum(0) \
            + (f_qr * w_qr.view(G, 1) * sq).sum(0)
        dRk[:, dd] = (f_kr * e_kr.view(G, 1) * ak[:, :, dd]).sum(0) \
            + (f_kr * f_kr * u_kr.view(G, 1) * torch.einsum("e,gec->gc", Rk[:, dd], gk)).sum(0) \
            + (f_kr * w_kr.view(G, 1) * sk).sum(0)
    dR = torch.cat([dRq, dRk, av], dim=0)
    # ---- gates: the dZ parts are dot products of accumulators, the u / w parts come from the forward statistics --
    var3 = 1.0 / (rstd3 * rstd3) - EPS                        # biased batch variance of the gated similarities
    sum_S = count * mean3                                     # sum S_x,  S_qr = f_qr rq
    sum_S2 = count * (var3 + mean3 * mean3)
    g_qr = (e_qr * A_qr / f_qr + u_qr * sum_S2[1] / f_qr + w_qr * sum_S[1] / f_qr).sum()
    g_kr = (e_kr * A_kr / f_kr + u_kr * sum_S2[2] / f_kr + w_kr * sum_S[2] / f_kr).sum()
    g_sv = (P * dPv).sum()
    g_sve = (P * dPe).sum()
    return dqkv, dR, torch.stack([g_qr, g_kr, g_sv, g_sve]), dgamma


@pytest.mark.parametrize("gp,L,training", [(2, 8, True), (4, 6, True), (8, 4, True), (2, 8, False)])
def test_single_sweep_backward_matches_autograd(gp, L, training):
    torch.manual_seed(gp * 100 + L)
    B, G = 3, 2
    dt = torch.float64
    qkv = torch.randn(B, G, 2 * gp, L, dtype=dt, requires_grad=True)
    R = (torch.randn(2 * gp, 2 * L - 1, dtype=dt) * 0.7).requires_grad_(True)
    gates = (torch.tensor([0.3, 0.45, 0.9, 0.2], dtype=dt) + 0.1).requires_grad_(True)
    gamma = (torch.rand(3 * G, dtype=dt) + 0.5).requires_grad_(True)
    beta = torch.randn(3 * G, dtype=dt, requires_grad=True)
    run = (torch.randn(3 * G, dtype=dt) * 0.1, torch.rand(3 * G, dtype=dt) + 0.5)
    stk, stats = core_forward(qkv, R, gates.unbind(0), gamma, beta, training, run)
    dstk = torch.randn_like(stk)
    (stk * dstk).sum().backward()
    if not training:
        stats["mean"], stats["rstd"] = run[0], torch.rsqrt(run[1] + EPS)
    with torch.no_grad():
        dqkv, dR, dgates, dgamma = single_sweep_backward(qkv, R, gates.unbind(0), gamma, stats, dstk, stk.detach(), training)
    tol = 1e-10
    assert (dqkv - qkv.grad).abs().max() < tol * max(1.0, qkv.grad.abs().max())
    assert (dR - R.grad).abs().max() < tol * max(1.0, R.grad.abs().max())
    assert (dgamma - gamma.grad).abs().max() < tol * max(1.0, gamma.grad.abs().max())
    assert beta.grad.abs().max() < 1e-9                      # sum dZ = 0: bn_similarity.bias has no gradient
    assert (dgates - gates.grad).abs().max() < tol * max(1.0, gates.grad.abs().max())     # (eval: u = w = 0)
