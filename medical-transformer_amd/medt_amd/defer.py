"""Host side of the deferred, grouped launches (include/medt_abi.h: medt_queue_*; csrc/defer.h).

While a StepQueue is `active()`, the layer entry points record weight / bias gradients, partial-slab reductions and the
statistics bookkeeping of the fused small-layer kernels instead of launching them; `flush()` issues everything recorded
as a few grouped launches.  The recorded jobs point into tensors of the Python call that recorded them (workspace,
inputs, saved activations, gradient outputs), so the queue holds a reference to each of them until the flush.
Only medt_amd.trainer.TrainStep uses this: it flushes after the forward pass and after the backward pass.
"""
from __future__ import annotations

import contextlib
import os

import torch

from . import _lib as L

ENABLED = os.environ.get("MEDT_DEFER", "1") != "0"
_current = None


class StepQueue:
    def __init__(self):
        self._h = L.lib().medt_queue_create()
        self._keep = []
        self._bound = set()

    def __del__(self):
        try:
            L.lib().medt_queue_destroy(self._h)
        except Exception:
            pass

    def bind_current_stream(self):
        s = torch.cuda.current_stream().cuda_stream
        if s not in self._bound:
            L.check(L.lib().medt_queue_bind(self._h, s), "medt_queue_bind")
            self._bound.add(s)

    def hold(self, *tensors):
        self._keep.extend(t for t in tensors if t is not None)

    def pending(self) -> int:
        return int(L.lib().medt_queue_pending(self._h))

    def flush(self):
        """Issue everything recorded so far on the current stream (all recording streams must have been joined into it)."""
        L.check(L.lib().medt_queue_flush(self._h, torch.cuda.current_stream().cuda_stream), "medt_queue_flush")
        self._keep.clear()

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
        try:
            yield self
            self.flush()
        finally:
            _current = None
            self._unbind_all()
            self._keep.clear()


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
