"""One training step of reference train.py:140,156-161 as a replayable HIP graph.

    output = model(X); loss = criterion(output, y); optimizer.zero_grad(); loss.backward(); optimizer.step()

At BASELINE.json's batch sizes every kernel of the step moves a few MB at most, so the step is bound by
launch latency, not by HBM or MFMA.  The whole forward + loss + backward + gradient packing
(+ the fused Adam when single-GPU) is therefore captured once into a hipGraph (torch.cuda.CUDAGraph drives
hipStreamBeginCapture; the kernels are enqueued by libmedt_hip.so on the capturing stream) and replayed per
step.  With several ranks the flat-bucket all-reduce runs between the replay and the Adam launch.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from .optim import FlatAdam
from .ops import cross_entropy


def _world():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def _distributed():
    """True when the step must leave the all-reduce (and Adam) outside the captured graph."""
    from .optim import FORCE_COLLECTIVES
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)


class TrainStep:
    def __init__(self, model, optimizer: FlatAdam, criterion=None, use_graph: bool = True, warmup: int = 3):
        self.model, self.opt = model, optimizer
        self.criterion = criterion if criterion is not None else cross_entropy
        self.use_graph = use_graph
        self.warmup = warmup
        self._graphs = {}          # signature -> (graph, static_x, static_y, static_loss, single)

    # ---- eager ------------------------------------------------------------
    def _eager(self, x, y):
        out = self.model(x)
        loss = self.criterion(out, y)
        self.opt.zero_grad()
        loss.backward()
        self.opt.pack_gradients()
        self.opt.allreduce()
        self.opt.apply(_world())
        return loss

    # ---- graph ------------------------------------------------------------
    def _signature(self, x, y):
        return (tuple(x.shape), tuple(y.shape), self.model.training,
                tuple(p.requires_grad for p in self.opt.params))

    def _capture(self, x, y):
        """Note: the warm-up steps are real optimisation steps on (x, y) (the same batch is then replayed)."""
        static_x, static_y = x.clone(), y.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                  # warm-up: allocator pools, FlatAdam groups, lazy inits
            for _ in range(self.warmup):
                self._eager(static_x, static_y)
        torch.cuda.current_stream().wait_stream(side)
        self.opt.zero_grad()
        graph = torch.cuda.CUDAGraph()
        single = not _distributed()
        with torch.cuda.graph(graph):
            out = self.model(static_x)
            loss = self.criterion(out, static_y)
            loss.backward()
            self.opt.pack_gradients()
            if single:
                self.opt.apply(1)
        return graph, static_x, static_y, loss.detach(), single

    def __call__(self, x, y):
        if not self.use_graph:
            return self._eager(x, y)
        sig = self._signature(x, y)
        entry = self._graphs.get(sig)
        if entry is None:                              # first call with this shape, or the gates were switched on
            if any(k[3] != sig[3] for k in self._graphs):          # requires_grad changed (train.py:169-171):
                self._graphs.clear()                                # graphs captured before are stale
            entry = self._graphs[sig] = self._capture(x, y)
        graph, static_x, static_y, static_loss, single = entry
        static_x.copy_(x)
        static_y.copy_(y)
        graph.replay()
        if not single:
            self.opt.allreduce()
            self.opt.apply(_world())
        return static_loss
