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
# Two gradient buckets for two-branch networks (MedT's global / local branch): the global branch's backward ends first, its
# bucket is all-reduced there, under the local branch's remaining backward chain.  MEDT_TWO_BUCKETS=0: one all-reduce per group
# after the whole backward (rounds 1-4).
TWO_BUCKETS = os.environ.get("MEDT_TWO_BUCKETS", "1") != "0"
EARLY_ENABLED = True   # TrainStep clears it while capturing a graph that must not contain collectives
_OPEN = None           # the FlatAdam whose step is open (between zero_grad() and pack_gradients()), for branch_done()


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


def branch_done(segment: int = 0):
    """Called by the backward where one branch of a two-branch network has finished (ops.ConvBlockFn, cfg.last_of_branch,
    after that stream's recorded weight-gradient jobs have been issued): its gradient bucket can go on the wire now."""
    if _OPEN is not None and TWO_BUCKETS:
        _OPEN.early_allreduce(segment)


class _Group:
    def __init__(self, params: List[torch.nn.Parameter], owner):
        dev = params[0].device
        # bucket layout: segment 0 (reduced early: the branch whose backward ends first + the trunk) in front, then the rest;
        # inside a segment the order of model.parameters().  Adam is elementwise: the layout changes no result.
        seg = owner.segment_of
        params = sorted(params, key=seg) if seg is not None else params          # (stable)
        self.params = params
        self.numel = sum(p.numel() for p in params)
        n0 = sum(p.numel() for p in params if seg is not None and seg(p) == 0)
        self.bounds = [(0, self.numel)] if n0 in (0, self.numel) else [(0, n0), (n0, self.numel)]
        self.seg_params = [[p for p in params if seg is None or (seg(p) == 0) == (k == 0)] for k in range(len(self.bounds))]
        self.reduced = [-1] * len(self.bounds)                # optimizer stamp of the step a segment was last all-reduced in
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
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, segment_of=None):
        """segment_of: parameter -> 0 (bucket that can be all-reduced as soon as branch_done() is called) or 1; None = one
        bucket.  TrainStep sets it from the model (set_segments_from_model) before the first step."""
        self.segment_of = segment_of
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
        global _OPEN
        self.stamp += 1
        self.step_open = True
        _OPEN = self
        for p in self.params:
            p.grad = None

    def set_segments_from_model(self, model):
        """Two-branch networks (medt_net: every module of the local branch is named *_p, reference axialnet.py:557-600): the
        global branch and the trunk (decoderf, adjust) form segment 0, the local branch segment 1.  Must be called before the
        first step (the flat layout is fixed when a group is adopted); single-branch networks keep one bucket."""
        if self.groups or not TWO_BUCKETS:
            return
        local = {id(p) for name, m in model.named_children() if name.endswith("_p") for p in m.parameters()}
        if local and len(local) < len(self.params):
            self.segment_of = lambda p: 1 if id(p) in local else 0

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
        global _OPEN
        self._adopt_new()
        self.step_open = False
        if _OPEN is self:
            _OPEN = None
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
                todo = [k for k in range(len(g.bounds)) if g.reduced[k] != self.stamp]
                if len(todo) == len(g.bounds):
                    dist.all_reduce(g.flat_g, op=dist.ReduceOp.SUM)          # nothing went out early: one collective
                else:
                    for k in todo:
                        dist.all_reduce(g.flat_g[g.bounds[k][0]:g.bounds[k][1]], op=dist.ReduceOp.SUM)
                g.reduced = [-1] * len(g.bounds)               # (a replayed graph calls this again without a zero_grad())

    def early_allreduce(self, segment: int = 0):
        """All-reduce one segment of every bucket NOW (on the current stream, in the middle of the backward pass), provided its
        gradients are complete: every parameter of the segment has had its slot written by a backward kernel of this step (no
        gradient still travelling through autograd's `.grad`, no first-step adoption pending).  Otherwise nothing happens and
        allreduce() covers the segment as before.  Every rank takes the same decision: it depends on the model only."""
        if not (collectives_needed() and self.step_open and EARLY_ENABLED):
            return
        # (parameters that join this step -- the gates at epoch 10 -- form a NEW group at pack_gradients(); the existing groups'
        #  layout does not change, and allreduce() covers whatever did not go out here)
        for g in self.groups:
            if len(g.bounds) < 2 or g.reduced[segment] == self.stamp:
                continue
            if any(p._medt_gslot.stamp != self.stamp or p.grad is not None for p in g.seg_params[segment]):
                continue
            lo, hi = g.bounds[segment]
            dist.all_reduce(g.flat_g[lo:hi], op=dist.ReduceOp.SUM)
            g.reduced[segment] = self.stamp

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
