#!/usr/bin/env python
"""Round 6 debugging aid: error statistics of conv_block (forward, backward-data) against float64 torch on the shapes MedT's global
branch runs at 2 / 4 images -- max and RMS error relative to the RMS of the reference, so that a 1e-6 systematic difference shows
(the unit tests' 2e-4 of the maximum does not).  Run with MEDT_CONV_THIN=0 / 3 and compare."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "medical-transformer_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import copy
import torch, torch.nn as nn
from medt_amd import ops
dev = torch.device("cuda:0")
cases = [(8, 128, 64, 2, True), (128, 8, 64, 2, True), (32, 16, 64, 2, False), (16, 16, 128, 2, False), (8, 128, 64, 4, True), (128, 8, 64, 4, True),
         (16, 16, 128, 4, False), (32, 16, 16, 32, False)]
for Cin, Cout, S, N, has_bn in cases:
    torch.manual_seed(Cin * 7 + Cout)
    conv = nn.Conv2d(Cin, Cout, 3, padding=1, bias=not has_bn)
    bn = nn.BatchNorm2d(Cout) if has_bn else None
    x = torch.randn(N, Cin, S, S).relu_() + 0.3          # (a mean: the centred sums of the statistics have something to cancel)
    dout = torch.randn(N, Cout, S, S)
    c64, b64 = copy.deepcopy(conv).double(), (copy.deepcopy(bn).double() if bn is not None else None)
    x64 = x.double().requires_grad_(True)
    z64 = c64(x64)
    y64 = torch.relu(b64(z64)) if bn is not None else z64
    (y64 * dout.double()).sum().backward()
    cd, bd = copy.deepcopy(conv).to(dev), (copy.deepcopy(bn).to(dev) if bn is not None else None)
    xd = x.to(dev).requires_grad_(True)
    y = ops.conv_block(xd, cd, bd, None, has_bn, True, 1)
    (y * dout.to(dev)).sum().backward()
    def st(a, b):
        a, b = a.double().cpu(), b.double()
        e = (a - b)
        return f"max {e.abs().max().item() / b.pow(2).mean().sqrt().item():.2e} rms {e.pow(2).mean().sqrt().item() / b.pow(2).mean().sqrt().item():.2e} mean {e.mean().item() / b.pow(2).mean().sqrt().item():+.2e}"
    print(f"{Cin:3d}->{Cout:3d} S{S} N{N} bn{int(has_bn)}: y [{st(y.detach(), y64.detach())}]  dx [{st(xd.grad, x64.grad)}]  dw [{st(cd.weight.grad, c64.weight.grad)}]"
          + (f"  dgamma [{st(bd.weight.grad, b64.weight.grad)}]" if bn is not None else ""))
