#!/usr/bin/env python
"""Round 6: the grouped weight-gradient kernel on ONE recorded job at a time (rocprofv3 --kernel-trace --stats around this script gives
conv_wgrad_mfma_grouped_kernel's launch time per shape): what a workgroup-step costs when nothing else is in the launch.
    rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python scripts/wgrad_ubench.py"""
import os, sys, copy
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "medical-transformer_amd"))
import torch, torch.nn as nn
from medt_amd import ops
from medt_amd.defer import StepQueue
from medt_amd.optim import FlatAdam

dev = torch.device("cuda:0")
CASES = [  # Cin, Cout, K, N, S, has_bn       (stride 1, pad K // 2)
    (16, 16, 3, 4, 128, False),    # decoderf: 768 blocks x 2 steps
    (128, 8, 3, 4, 64, True),      # conv3: 288 blocks x 8 steps
    (8, 128, 3, 4, 64, True),      # conv2: 128 blocks x 8 steps
    (16, 32, 1, 4, 64, True),      # a qkv_transform of layer1: 16 blocks x 8 steps
    (16, 2, 1, 4, 128, False),     # adjust: 256 blocks x 2 steps
    (32, 64, 1, 64, 16, True),     # a local 1x1: 32 blocks x 8 steps
]
reps = int(os.environ.get("REPS", "30"))
for Cin, Cout, K, N, S, has_bn in CASES:
    torch.manual_seed(1)
    conv = nn.Conv2d(Cin, Cout, K, stride=1, padding=K // 2, bias=False).to(dev)
    bn = nn.BatchNorm2d(Cout).to(dev) if has_bn else None
    params = [conv.weight] + ([bn.weight, bn.bias] if bn is not None else [])
    opt = FlatAdam(params, lr=0.0)
    x = torch.randn(N, Cin, S, S, device=dev)
    dd = torch.randn(N, Cout, S, S, device=dev)

    def fwd_bwd(q=None):
        opt.zero_grad()
        y = ops.conv_block(x, conv, bn, None, True, True, 1)
        if q is not None:
            q.flush()
        (y * dd).sum().backward()
    fwd_bwd()
    opt.pack_gradients()
    q = StepQueue()
    torch.cuda.synchronize()
    # marker launch so the trace can be split per case: a fill of Cin*1000+Cout elements
    torch.zeros(Cin * 1000 + Cout + K, device=dev).add_(1.0)
    for _ in range(reps):
        with q.active():
            fwd_bwd(q)
    torch.cuda.synchronize()
    print("case", Cin, Cout, K, N, S, has_bn, "done", flush=True)
