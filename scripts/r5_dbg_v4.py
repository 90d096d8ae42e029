"""GPU debugging of conv_wgrad_k1v4_body32: error pattern of the recorded weight gradient per (o, c) for a few shapes."""
import sys, os, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "medical-transformer_amd"), ROOT, os.path.join(ROOT, "tests")]
import torch, torch.nn as nn
from medt_amd import ops
from medt_amd.defer import StepQueue
from medt_amd.optim import FlatAdam
dev = torch.device("cuda:0")
for case in ((16, 32, 1, 4, 8, 2), (72, 40, 0, 2, 10, 1), (16, 32, 1, 16, 32, 1), (32, 16, 1, 2, 16, 1), (16, 16, 0, 1, 16, 1), (16, 16, 0, 1, 8, 1), (32, 32, 0, 1, 16, 1)):
    Cin, Cout, has_bn, N, S, groups = case
    torch.manual_seed(Cin * 100 + Cout + 1 + S)
    conv = nn.Conv2d(Cin, Cout, 1, bias=False).to(dev)
    bn = nn.BatchNorm2d(Cout).to(dev) if has_bn else None
    x = torch.randn(N, Cin, S, S, device=dev); dd = torch.randn(N, Cout, S, S, device=dev)
    opt = FlatAdam([conv.weight] + ([bn.weight, bn.bias] if bn else []), lr=0.0)
    def fb(q=None):
        opt.zero_grad()
        y = ops.conv_block(x, conv, bn, None, True, True, groups)
        if q is not None: q.flush()
        (y * dd).sum().backward()
    fb(); opt.pack_gradients(); torch.cuda.synchronize()
    imm = conv.weight.grad.detach().clone().reshape(Cout, Cin)
    q = StepQueue()
    with q.active():
        fb(q)
    opt.pack_gradients(); torch.cuda.synchronize()
    g = conv.weight.grad.detach().reshape(Cout, Cin)
    err = (g - imm).abs() / imm.abs().max()
    bad = (err > 1e-4)
    print(case, "max rel err %.3e" % err.max().item(), "bad %d of %d" % (bad.sum().item(), bad.numel()),
          "bad rows(o):", sorted(set(bad.nonzero()[:, 0].tolist()))[:40], "bad cols(c):", sorted(set(bad.nonzero()[:, 1].tolist()))[:40], flush=True)
    if bad.any():
        o, c = bad.nonzero()[0].tolist()
        print("   e.g. (o=%d,c=%d): recorded %.6f immediate %.6f ratio %.4f" % (o, c, g[o, c].item(), imm[o, c].item(), g[o, c].item() / imm[o, c].item()))
