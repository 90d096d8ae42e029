"""Flat-buffer Adam: the optimizer step of reference train.py:111-112,161 as ONE kernel.

torch.optim.Adam(list(model.parameters()), lr, weight_decay=1e-5) semantics (coupled L2 decay, bias
correction, parameters that never receive a gradient are left untouched) over contiguous fp32 buffers:
parameters become views of one flat tensor, their gradients are packed into a matching flat tensor that
doubles as the data-parallel all-reduce bucket (dp.py), and medt_adam_step updates everything in one
launch with the step counter kept on the device (hipGraph-replayable).

Parameters join a flat group the first time they show up with a gradient.  The gates
(f_qr/f_kr/f_sve/f_sv, requires_grad=False until train.py:169-171 flips them at epoch 10) therefore
form a second group with its own step counter -- exactly torch.optim.Adam's per-parameter `step`.
"""
from __future__ import annotations

from typing import List

import torch
import torch.distributed as dist

from . import _lib as L


import os

FORCE_COLLECTIVES = os.environ.get("MEDT_FORCE_DIST") == "1"      # run the all-reduce even with one rank (tests)


class _Group:
    def __init__(self, params: List[torch.nn.Parameter]):
        dev = params[0].device
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.flat_p = torch.empty(self.numel, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        self.state = torch.zeros(3, device=dev, dtype=torch.float32)        # [step, 1-b1^t, 1-b2^t]
        self.gviews = []
        off = 0
        for p in params:
            n = p.numel()
            view = self.flat_p[off:off + n].view(p.shape)
            view.copy_(p.data)
            p.data = view                                     # the module now reads the flat buffer
            self.gviews.append(self.flat_g[off:off + n].view(p.shape))
            off += n


class FlatAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params]
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.groups: List[_Group] = []
        self._member = set()

    def zero_grad(self, set_to_none: bool = True):
        for p in self.params:
            p.grad = None

    def _adopt_new(self):
        new = [p for p in self.params if p.grad is not None and id(p) not in self._member]
        if new:
            for p in new:
                if not p.is_cuda or p.dtype != torch.float32:
                    raise L.MedtError("FlatAdam: float32 parameters on the GPU expected")
                self._member.add(id(p))
            self.groups.append(_Group(new))

    def pack_gradients(self):
        """Copy the autograd-produced gradients into the flat buckets (a few multi-tensor launches)."""
        self._adopt_new()
        for g in self.groups:
            grads = [p.grad for p in g.params]
            if any(gr is None for gr in grads):
                raise L.MedtError("FlatAdam: a parameter that used to receive gradients did not this step")
            torch._foreach_copy_(g.gviews, grads)

    def allreduce(self):
        """Sum the flat buckets over ranks (the 1/world factor is folded into the Adam kernel)."""
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES):
            for g in self.groups:
                dist.all_reduce(g.flat_g, op=dist.ReduceOp.SUM)

    def apply(self, world: int = 1):
        lib = L.lib()
        stream = torch.cuda.current_stream().cuda_stream
        for g in self.groups:
            L.check(lib.medt_adam_step(g.flat_p.data_ptr(), g.flat_g.data_ptr(), g.exp_avg.data_ptr(),
                                       g.exp_avg_sq.data_ptr(), g.state.data_ptr(), g.numel, self.lr, self.betas[0],
                                       self.betas[1], self.eps, self.weight_decay, 1.0 / world, stream),
                    "medt_adam_step")

    def step(self):
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.pack_gradients()
        self.allreduce()
        self.apply(world)

    # checkpointing parity with torch.optim.Adam is out of scope: the reference never saves optimizer state
    # (train.py:216-217 saves model.state_dict() only).
