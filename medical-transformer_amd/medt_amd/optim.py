"""Flat-buffer Adam: the optimizer step of reference train.py:111-112,161 as ONE kernel.

torch.optim.Adam(list(model.parameters()), lr, weight_decay=1e-5) semantics (coupled L2 decay, bias
correction, parameters that never receive a gradient are left untouched) over contiguous fp32 buffers:
parameters become views of one flat tensor and own a *gradient slot* -- a view of a matching flat
gradient tensor that doubles as the data-parallel all-reduce bucket.  The backward kernels of
medt_amd.ops / medt_amd.axial write parameter gradients straight into those slots (no autograd
AccumulateGrad copies, no packing pass); medt_adam_step then updates everything in one launch with the
step counter kept on the device (hipGraph-replayable).

Parameters join a flat group the first time they show up with a gradient (that first gradient comes
through autograd's ordinary `.grad` and is copied in once).  The gates (f_qr/f_kr/f_sve/f_sv,
requires_grad=False until train.py:169-171 flips them at epoch 10) therefore form a second group with
its own step counter -- exactly torch.optim.Adam's per-parameter `step`.  Membership is a property of
the model, not of the data, so every data-parallel rank builds identical buckets.
"""
from __future__ import annotations

import os
from typing import List

import torch
import torch.distributed as dist

from . import _lib as L

FORCE_COLLECTIVES = os.environ.get("MEDT_FORCE_DIST") == "1"      # run the all-reduce even with one rank (tests)


def collectives_needed() -> bool:
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)


class GradSlot:
    """Where a parameter's gradient lives: a view into its group's flat gradient bucket.

    `stamp` records the optimizer step in which the slot was last written, so that a parameter used twice
    in one forward (never the case in the reference's networks) still accumulates instead of overwriting."""
    __slots__ = ("view", "owner", "stamp")

    def __init__(self, view, owner):
        self.view, self.owner, self.stamp = view, owner, -1


def grad_slot(p):
    """The slot of parameter `p` (looked up in forward), or None (not adopted by a FlatAdam / frozen / not a leaf)."""
    s = getattr(p, "_medt_gslot", None)
    if s is None or not p.requires_grad:
        return None
    return s


def live(slot):
    """The slot if its owner has a step open, else None (asked in backward).

    Protocol: slots are written only between FlatAdam.zero_grad() and FlatAdam.pack_gradients() (the first thing
    FlatAdam.step() does).  A backward outside that window -- torch.autograd.grad, model.zero_grad() followed by a
    hand-written loop, a second optimizer -- gets None here, and its parameter gradients travel through autograd's
    ordinary `.grad` accumulation exactly as they would without a FlatAdam."""
    return slot if (slot is not None and slot.owner.step_open) else None


def claim(slot: GradSlot):
    """-> (tensor to write the gradient into, direct).  direct=False: the slot already holds this step's gradient
    of another use of the parameter; the caller writes a temporary and calls `accumulate`."""
    if slot.stamp != slot.owner.stamp:
        slot.stamp = slot.owner.stamp
        return slot.view, True
    return torch.empty_like(slot.view), False


def accumulate(slot: GradSlot, tmp):
    slot.view.add_(tmp)


class _Group:
    def __init__(self, params: List[torch.nn.Parameter], owner):
        dev = params[0].device
        self.params = params
        self.numel = sum(p.numel() for p in params)
        self.flat_p = torch.empty(self.numel, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(self.numel, device=dev, dtype=torch.float32)
        self.state = torch.zeros(3, device=dev, dtype=torch.float32)        # [step, 1-b1^t, 1-b2^t]
        off = 0
        for p in params:
            n = p.numel()
            view = self.flat_p[off:off + n].view(p.shape)
            view.copy_(p.data)
            p.data = view                                     # the module now reads the flat buffer
            gview = self.flat_g[off:off + n].view(p.shape)
            gview.copy_(p.grad)                               # the gradient of the adopting step
            p.grad = gview                                    # .grad IS the slot from now on
            p._medt_gslot = GradSlot(gview, owner)
            p._medt_gslot.stamp = owner.stamp
            off += n

    def tensors(self):
        return (self.flat_p, self.exp_avg, self.exp_avg_sq, self.state)


class FlatAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = [p for p in params]
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.groups: List[_Group] = []
        self._member = set()
        self.stamp = 0
        self.step_open = False         # True between zero_grad() and pack_gradients(): see grad_slot

    # ---- gradient bookkeeping ------------------------------------------------
    def zero_grad(self, set_to_none: bool = True):
        """Start a new step: `.grad` of every parameter is reset to None, as torch.optim does.  The slot-aware backward
        kernels write into the slots; pack_gradients() then points `.grad` back at them."""
        self.stamp += 1
        self.step_open = True
        for p in self.params:
            p.grad = None

    def _adopt_new(self):
        new = [p for p in self.params if p.grad is not None and id(p) not in self._member]
        if new:
            for p in new:
                if p.dtype != torch.float32:
                    raise L.MedtError("FlatAdam: float32 parameters expected")
                self._member.add(id(p))
            self.groups.append(_Group(new, self))

    def pack_gradients(self):
        """Make the flat buckets hold this step's gradients.

        Parameters whose backward kernel wrote its slot directly need nothing (the normal case: zero launches);
        parameters that received their first gradient are adopted into a new group; gradients that arrived through
        plain autograd (a module that is not slot-aware) are copied into their slot."""
        if not self.step_open:
            # No FlatAdam.zero_grad() since the last pack: the backward (if any) ran outside the slot window -- e.g.
            # model.zero_grad() + a hand-written loop -- so `live()` kept the kernels off the slots and every gradient
            # travelled through autograd's `.grad`.  The slots still carry the PREVIOUS step's stamp and contents; a new
            # stamp makes them stale, so a fresh `.grad` tensor below REPLACES the slot instead of being added to it.
            self.stamp += 1
        self._adopt_new()
        self.step_open = False
        for g in self.groups:
            src, dst = [], []
            for p in g.params:
                s = p._medt_gslot
                if s.stamp == self.stamp:
                    if p.grad is not None and p.grad is not s.view:          # both routes contributed
                        s.view.add_(p.grad)
                elif p.grad is not None:
                    if p.grad is not s.view:         # (`.grad` still being the slot = autograd accumulated in place)
                        dst.append(s.view)
                        src.append(p.grad)
                    s.stamp = self.stamp
                elif p.requires_grad:
                    raise L.MedtError("FlatAdam: a parameter that used to receive gradients did not this step")
                else:
                    # torch.optim.Adam skips a parameter without a gradient entirely (no decay, no momentum step); the
                    # single flat kernel cannot skip a range, and the reference never freezes a trained parameter
                    # (train.py:169-171 only ever unfreezes the gates)
                    raise L.MedtError("FlatAdam: a parameter was frozen after it had been trained; rebuild the optimizer")
                p.grad = s.view
            if src:
                torch._foreach_copy_(dst, src)

    def signature(self):
        """What a captured step depends on: group membership and which parameters are trainable."""
        return (tuple(g.numel for g in self.groups), tuple(p.requires_grad for p in self.params))

    # ---- data parallel -----------------------------------------------------------
    def allreduce(self):
        """Sum the flat buckets over ranks (the 1/world factor is folded into the Adam kernel)."""
        if collectives_needed():
            for g in self.groups:
                dist.all_reduce(g.flat_g, op=dist.ReduceOp.SUM)

    # ---- update ----------------------------------------------------------------------
    def _launch_adam(self, g: _Group, gscale: float):
        if not g.flat_p.is_cuda:
            raise L.MedtError("FlatAdam: parameters must live on the GPU (medt_adam_step is a HIP kernel; there is no "
                              "CPU fallback)")
        lib = L.lib()
        stream = torch.cuda.current_stream().cuda_stream
        L.check(lib.medt_adam_step(g.flat_p.data_ptr(), g.flat_g.data_ptr(), g.exp_avg.data_ptr(),
                                   g.exp_avg_sq.data_ptr(), g.state.data_ptr(), g.numel, self.lr, self.betas[0],
                                   self.betas[1], self.eps, self.weight_decay, gscale, stream), "medt_adam_step")

    def apply(self, world: int = 1):
        for g in self.groups:
            self._launch_adam(g, 1.0 / world)

    def step(self):
        world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.pack_gradients()
        self.allreduce()
        self.apply(world)

    # ---- snapshot / restore (hipGraph warm-up must not advance the optimisation, trainer.py) --------------
    def snapshot(self):
        return [tuple(t.clone() for t in g.tensors()) for g in self.groups]

    def restore(self, snap):
        """Groups that existed at snapshot time get their values back; groups adopted since are reset to a fresh
        optimizer state (their parameters are restored by the caller, who snapshots the model)."""
        for i, g in enumerate(self.groups):
            if i < len(snap):
                for t, s in zip(g.tensors(), snap[i]):
                    t.copy_(s)
            else:
                g.exp_avg.zero_()
                g.exp_avg_sq.zero_()
                g.state.zero_()

    # checkpointing parity with torch.optim.Adam is out of scope: the reference never saves optimizer state
    # (train.py:216-217 saves model.state_dict() only).
