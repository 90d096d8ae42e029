"""Host side of the deferred, grouped launches (include/medt_abi.h: medt_queue_*; csrc/defer.h).

While a StepQueue is `active()`, the layer entry points record weight / bias gradients, partial-slab reductions and the
statistics bookkeeping of the fused small-layer kernels instead of launching them; `flush()` issues everything recorded
as a few grouped launches.  The recorded jobs point into tensors of the Python call that recorded them (workspace,
inputs, saved activations, gradient outputs), so the queue holds a reference to each of them until the flush.
medt_amd.trainer.TrainStep flushes after the forward pass and after the backward pass; InferStep (no backward) drops the
bookkeeping of an eval-mode forward altogether (`drop`).
"""
from __future__ import annotations

import contextlib
import os

import torch

from . import _lib as L

ENABLED = os.environ.get("MEDT_DEFER", "1") != "0"
# one queue per recording stream: the stream whose pass ends first (MedT's global branch) issues its own grouped
# launches right there, under the other branch's latency-bound chain, instead of after the join (net.medt_forward)
SPLIT = os.environ.get("MEDT_SPLIT_FLUSH", "1") != "0"
# the pass's LAST flush (the longer branch's recorded jobs, behind the step's critical chain) runs its dedicated MFMA weight-gradient
# launches on the other branch's stream, which is idle by then (medt_queue_flush2; net.medt_forward names the stream)
TAIL_FORK = os.environ.get("MEDT_TAIL_FORK", "1") != "0"
_current = None
_aux = None            # torch.cuda.Stream of the other branch (two-branch networks), or None


def set_aux_stream(stream):
    global _aux
    _aux = stream


class StepQueue:
    def __init__(self):
        self._handles = {}             # stream -> queue handle (one shared handle under key None when not SPLIT)
        self._keep = {}                # handle -> tensors the recorded jobs point into
        self._bound = set()
        self.issued = 0                # jobs handed to grouped launches so far (tests: the forward flushes per branch)
        self.drop = False              # True: every flush discards instead (InferStep in eval mode: bookkeeping nobody reads)

    def __del__(self):
        try:
            for h in self._handles.values():
                L.lib().medt_queue_destroy(h)
        except Exception:
            pass

    def _handle(self, s):
        key = s if SPLIT else None
        h = self._handles.get(key)
        if h is None:
            h = self._handles[key] = L.lib().medt_queue_create()
            self._keep[h] = []
        return h

    def bind_current_stream(self):
        s = torch.cuda.current_stream().cuda_stream
        if s not in self._bound:
            L.check(L.lib().medt_queue_bind(self._handle(s), s), "medt_queue_bind")
            self._bound.add(s)

    def hold(self, *tensors):
        h = self._handle(torch.cuda.current_stream().cuda_stream)
        self._keep[h].extend(t for t in tensors if t is not None)

    def pending(self) -> int:
        return sum(int(L.lib().medt_queue_pending(h)) for h in self._handles.values())

    def flush(self):
        """Issue everything recorded so far on the current stream (all recording streams must have been joined into it)."""
        if self.drop:
            return self.discard()
        cur = torch.cuda.current_stream().cuda_stream
        aux = _aux.cuda_stream if (TAIL_FORK and _aux is not None and torch.is_grad_enabled()) else None
        for h in self._handles.values():
            self.issued += int(L.lib().medt_queue_pending(h))
            if aux is not None and aux != cur:
                L.check(L.lib().medt_queue_flush2(h, cur, aux), "medt_queue_flush2")
            else:
                L.check(L.lib().medt_queue_flush(h, cur), "medt_queue_flush")
            self._keep[h].clear()

    def flush_current_stream(self):
        """Issue what was recorded ON the current stream, on it (no other stream's work is waited for)."""
        if not SPLIT:
            return
        cur = torch.cuda.current_stream().cuda_stream
        h = self._handles.get(cur)
        if h is not None and self.drop:
            L.lib().medt_queue_discard(h)
            self._keep[h].clear()
        elif h is not None:
            self.issued += int(L.lib().medt_queue_pending(h))
            L.check(L.lib().medt_queue_flush(h, cur), "medt_queue_flush")
            self._keep[h].clear()

    def discard(self):
        """Drop everything recorded without launching it (the step raised: the tensors the jobs point into are about
        to be released, so a later flush would write through dangling pointers)."""
        lib = L.lib()
        for h in self._handles.values():
            lib.medt_queue_discard(h)
            self._keep[h].clear()

    def _unbind_all(self):
        lib = L.lib()
        for s in self._bound:
            lib.medt_queue_bind(None, s)
        self._bound.clear()

    @contextlib.contextmanager
    def active(self):
        global _current
        if not ENABLED or _current is not None:
            yield self
            return
        _current = self
        ok = False
        try:
            yield self
            self.flush()
            ok = True
        finally:
            _current = None
            self._unbind_all()
            if not ok:             # the body (or the flush) raised: nothing recorded may run later
                self.discard()
            for k in self._keep.values():
                k.clear()


def flush_current_stream() -> bool:
    """Called where one branch's backward pass ends (ops.ConvBlockFn, cfg.last_of_branch).  True: everything the branch
    recorded has been issued on the current stream (or nothing is ever recorded: immediate launches) -- its parameter gradients
    are complete in stream order.  False: one shared queue (MEDT_SPLIT_FLUSH=0), flushed at the end of the pass."""
    if _current is None:
        return True
    if not SPLIT:
        return False
    _current.flush_current_stream()
    return True


def recording(allow: bool = True):
    """The active queue, with the current stream bound to it (call right before a library entry point), or None.

    allow=False: this call must launch immediately -- e.g. a backward whose parameter gradients go back through autograd
    (which copies them as soon as the function returns) instead of into persistent FlatAdam slots; the stream is
    unbound for the call."""
    q = _current
    if q is None:
        return None
    if allow:
        q.bind_current_stream()
        return q
    s = torch.cuda.current_stream().cuda_stream
    if s in q._bound:
        L.lib().medt_queue_bind(None, s)
        q._bound.discard(s)
    return None
