"""Does computing BatchNorm batch variance as E[x^2] - E[x]^2 from float32 sums (what the HIP producers' partial sums
amount to; finalised in float64) add to the float32 noise of the whole-network training-mode gradients?

Runs the CPU oracle (test infrastructure) three ways on the same inputs: float64 (truth), float32 with torch's own
variance, float32 with the raw-moment variance.  Prints, per model, the distribution over gradient tensors of
|g32 - g64| for both float32 variants.   python scripts/bn_variance_noise.py [model] [S] [N]
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from oracle import medt_oracle as O          # noqa: E402
import helpers as H                          # noqa: E402


def run(name, S, N, seed, dtype, raw_moments):
    st = H.seeded_state(name, S, seed)
    x, y = H.seeded_input(seed + 1, N, 3, S)
    ost = O.clone_state(st, dtype, requires_grad=True)
    orig = O.batch_norm
    if raw_moments:
        def bn(xx, stt, prefix, training, bn_groups=1):
            if not training:
                return orig(xx, stt, prefix, training, bn_groups)
            w, b = stt[prefix + ".weight"], stt[prefix + ".bias"]
            C = xx.shape[1]
            shape = [1] * xx.dim()
            shape[1] = C
            outs = []
            for xg in xx.chunk(bn_groups, dim=0):
                dims = [d for d in range(xg.dim()) if d != 1]
                n = xg.numel() // C
                s1 = xg.sum(dim=dims)                         # float32 sums (pairwise inside torch: kinder than a GPU tree)
                s2 = (xg * xg).sum(dim=dims)
                mean = (s1.double() / n)
                var = (s2.double() / n - mean * mean).clamp_min(0)
                mean, var = mean.to(xg.dtype), var.to(xg.dtype)
                # straight-through for autograd: value from the raw moments, gradient of the exact formula
                em = xg.mean(dim=dims)
                ev = xg.var(dim=dims, unbiased=False)
                mean = em + (mean - em).detach()
                var = ev + (var - ev).detach()
                outs.append((xg - mean.view(shape)) * torch.rsqrt(var.view(shape) + O.BN_EPS) * w.view(shape) + b.view(shape))
            return torch.cat(outs, 0) if bn_groups > 1 else outs[0]
        O.batch_norm = bn
    try:
        out = O.forward(name, x.to(dtype), ost, True)
        O.log_nll_loss(out, y).backward()
    finally:
        O.batch_norm = orig
    return {k: v.grad.double() for k, v in ost.items() if v.grad is not None}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "gatedaxialunet"
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    N = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    torch.set_num_threads(8)
    g64 = run(name, S, N, 101, torch.float64, False)
    g32 = run(name, S, N, 101, torch.float32, False)
    g32r = run(name, S, N, 101, torch.float32, True)
    ratios = []
    for k in g64:
        e1 = (g32[k] - g64[k]).norm().item()
        e2 = (g32r[k] - g64[k]).norm().item()
        ratios.append((e2 / max(e1, 1e-30), k, e1, e2))
    ratios.sort()
    r = [v[0] for v in ratios]
    print(f"{name} S={S} N={N}: ||g32_rawmoments - g64|| / ||g32_torchvar - g64|| per tensor: median {r[len(r)//2]:.2f}, "
          f"90% {r[int(0.9*len(r))]:.2f}, max {r[-1]:.2f} ({ratios[-1][1]})")


if __name__ == "__main__":
    main()
