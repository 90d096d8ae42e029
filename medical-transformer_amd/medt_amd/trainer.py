"""One training step of reference train.py:140,156-161 as a replayable HIP graph.

    output = model(X); loss = criterion(output, y); optimizer.zero_grad(); loss.backward(); optimizer.step()

At BASELINE.json's batch sizes every kernel of the step moves a few MB at most, so the step is bound by
launch latency, not by HBM or MFMA.  The whole forward + loss + backward + gradient packing
(+ the fused Adam when single-GPU) is therefore captured once into a hipGraph (torch.cuda.CUDAGraph drives
hipStreamBeginCapture; the kernels are enqueued by libmedt_hip.so on the capturing stream) and replayed per
step.  With several ranks the flat-bucket all-reduce (RCCL) and the Adam launch behind it are graph nodes too.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from .defer import StepQueue
from .optim import FlatAdam
from .ops import cross_entropy


def _world():
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def _distributed():
    """True when the step must leave the all-reduce (and Adam) outside the captured graph."""
    from .optim import collectives_needed
    return collectives_needed()


class TrainStep:
    def __init__(self, model, optimizer: FlatAdam, criterion=None, use_graph: bool = True, warmup: int = 3):
        if warmup < 1:
            raise ValueError("TrainStep: at least one warm-up step is needed before capture (FlatAdam adopts the "
                             "parameters and re-points their storage on the first step)")
        self.model, self.opt = model, optimizer
        if hasattr(optimizer, "set_segments_from_model"):
            optimizer.set_segments_from_model(model)          # two-branch networks: two gradient buckets (optim.TWO_BUCKETS)
        self.criterion = criterion if criterion is not None else cross_entropy
        self.use_graph = use_graph
        self.warmup = warmup
        self._graphs = {}          # signature -> (graph, static_x, static_y, static_loss, single)
        self._queue = None         # deferred, grouped launches (medt_amd.defer); created on first use (needs the GPU library)
        self._ce_out = None        # [loss, counted pixels, out-of-range targets] of the last step (medt cross_entropy)
        self.collective_in_graph = False                      # set by capture: the all-reduce + Adam are graph nodes

    # ---- eager ------------------------------------------------------------
    def _fwd_bwd(self, x, y):
        """forward -> loss -> backward.  On the GPU the work no later layer waits for (weight / bias gradients and their
        slab reductions, statistics bookkeeping of the fused small-layer kernels) is recorded and issued as grouped
        launches at the end of each pass: before backward reads the saved statistics, before the optimizer reads the
        gradients."""
        if not x.is_cuda:
            out = self.model(x)
            loss = self.criterion(out, y)
            self.opt.zero_grad()
            loss.backward()
        else:
            if self._queue is None:
                self._queue = StepQueue()
            with self._queue.active() as q:
                out = self.model(x)
                try:
                    loss = self.criterion(out, y)
                except BaseException:
                    # e.g. class indices outside [0, K): the forward pass is complete, so its recorded bookkeeping
                    # (saved statistics, running-stat updates) is issued like the reference's eager forward would
                    # have; whatever else is pending when a step dies is dropped by StepQueue.active()
                    q.flush()
                    raise
                q.flush()
                self.opt.zero_grad()
                # (a persistent ones tensor handed to backward() would save autograd's fill launch on the chain, ~4 us -- but every
                #  eager forward AFTER a step captured that way ran 2x slower, 1.42 vs 0.70 ms/image: measured, not understood)
                loss.backward()
        self.opt.pack_gradients()
        return loss

    def _eager(self, x, y):
        loss = self._fwd_bwd(x, y)
        self.opt.allreduce()
        self.opt.apply(_world())
        return loss

    def check_targets(self):
        """Raise if the last replayed step saw class indices outside [0, K) (what F.cross_entropy raises on).  One
        host sync: call it where the loss is read anyway (train.py does, next to loss.item())."""
        if self._ce_out is not None:
            from .ops import raise_on_bad_targets
            raise_on_bad_targets(self._ce_out, -1)

    # ---- graph ------------------------------------------------------------
    def _signature(self, x, y):
        return (tuple(x.shape), tuple(y.shape), self.model.training, self.opt.signature())

    def _snapshot(self):
        tensors = list(self.model.parameters()) + list(self.model.buffers())
        return [t.detach().clone() for t in tensors], self.opt.snapshot()

    def _restore(self, snap):
        values, opt_snap = snap
        tensors = list(self.model.parameters()) + list(self.model.buffers())
        with torch.no_grad():
            for t, v in zip(tensors, values):
                t.copy_(v)
        self.opt.restore(opt_snap)

    def _capture(self, x, y):
        """Warm up (allocator pools, FlatAdam adoption, lazy inits), then capture.  The warm-up steps run on the real
        batch but their effect -- weights, Adam moments and step counters, BatchNorm running statistics,
        num_batches_tracked -- is rolled back before capture, so capture + replay performs exactly the one update the
        reference's loop (train.py:159-161) performs for this batch."""
        from .data import GPU_CAPTURE_LOCK
        with GPU_CAPTURE_LOCK:         # no prefetch-thread runtime calls while the stream is capturing
            return self._capture_locked(x, y)

    def _capture_locked(self, x, y):
        static_x, static_y = x.clone(), y.clone()
        snap = self._snapshot()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(self.warmup):
                self._eager(static_x, static_y)
            self._restore(snap)
        torch.cuda.current_stream().wait_stream(side)
        single = not _distributed()
        # Data parallel: the flat-bucket all-reduce (RCCL through torch's process group -- its kernels are capturable once
        # the communicator exists, which the eager warm-up steps above guarantee) and the Adam launch behind it are captured
        # INTO the graph: one replay per step, no host round trip between backward, collective and update.
        # MEDT_GRAPH_COLLECTIVE=0, or a process group whose collectives cannot be captured (gloo), leaves them outside.
        in_graph = single or (os.environ.get("MEDT_GRAPH_COLLECTIVE", "1") != "0" and self._collective_capturable()
                              and self._collective_capture_probe(static_x.device))
        from . import optim as OPT
        for attempt in (0, 1):
            graph = torch.cuda.CUDAGraph()
            OPT.EARLY_ENABLED = in_graph                      # no early bucket all-reduce inside a graph without collectives
            try:
                # thread_local: runtime calls of OTHER host threads (a data-loader / pin-memory thread of the caller's own)
                # do not invalidate this capture; the prefetcher of medt_amd.data additionally holds GPU_CAPTURE_LOCK
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    loss = self._fwd_bwd(static_x, static_y)
                    if in_graph:
                        if not single:
                            self.opt.allreduce()
                        self.opt.apply(_world())
                break
            except RuntimeError:
                if single or not in_graph or attempt:
                    raise
                in_graph = False                              # the collective refused capture: keep it outside the graph
                torch.cuda.synchronize()
                self._restore(snap)
            finally:
                OPT.EARLY_ENABLED = True
        self.collective_in_graph = in_graph and not single
        return graph, static_x, static_y, loss.detach(), in_graph, getattr(loss, "_medt_ce_out", None)

    @staticmethod
    def _collective_capture_probe(device):
        """Capture ONE tiny all-reduce in a throw-away graph: a process group whose collectives refuse stream capture raises here,
        in the capturing thread, where the capture can be unwound cleanly.  (Since round 5 the step's first collective is issued
        from inside the backward pass -- optim.branch_done, autograd's engine thread, other streams still forked: a refusal there
        cannot be unwound, the capture stays active and the next capture_begin fails.)"""
        t = torch.zeros(8, device=device)
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                dist.all_reduce(t, op=dist.ReduceOp.SUM)
        except RuntimeError:
            torch.cuda.synchronize()
            return False
        return True

    @staticmethod
    def _collective_capturable():
        try:
            return dist.get_backend() == "nccl"               # RCCL on ROCm
        except Exception:
            return False

    def __call__(self, x, y):
        if not self.use_graph:
            return self._eager(x, y)
        sig = self._signature(x, y)
        entry = self._graphs.get(sig)
        if entry is None:                              # first call with this shape, or the gates were switched on
            if any(k[3] != sig[3] for k in self._graphs):          # trainable set changed (train.py:169-171):
                self._graphs.clear()                                # graphs captured before are stale
            entry = self._capture(x, y)
            self._graphs[self._signature(x, y)] = entry             # (the warm-up may have adopted a new group)
        graph, static_x, static_y, static_loss, single, self._ce_out = entry
        static_x.copy_(x)
        static_y.copy_(y)
        graph.replay()
        if not single:
            self.opt.allreduce()
            self.opt.apply(_world())
        return static_loss


class InferStep:
    """`with torch.no_grad(): y = model(x)` -- the loop body of reference test.py:106-119 and of train.py's periodic
    validation (train.py:174-184) -- as a replayable HIP graph per input shape, optionally with the device-side
    segmentation counts (ops.seg_counts) behind it.

    An eager forward is ~110 dependent launches of 5-25 us kernels issued from Python: it is host-bound (0.7 ms/image at
    batch 4 -- longer than the whole replayed TRAINING step that contains the same forward).  Replayed, the forward costs
    what its two kernel chains cost.  Semantics kept:

      * model.eval(): BatchNorm is a per-channel affine of the running statistics.  The statistics bookkeeping the
        fused small-layer kernels record (saved mean / rstd blocks that only a backward pass reads) is DROPPED from the
        graph (StepQueue.discard) -- nothing in a no-grad forward reads it;
      * model.train() under no_grad (what the reference's validation loop does, train.py:174-184: the model is never
        switched to eval there): batch statistics, and every replay applies the running-statistics update and bumps
        num_batches_tracked once, like the eager call; the recorded bookkeeping is flushed inside the graph.

    The returned tensors are the graph's static outputs: valid until the next call with the same signature (clone to
    keep).  Parameters and buffers are read through their storage at replay time, so optimizer updates between replays
    are seen; re-pointing parameter storage (FlatAdam adoption on the first training step, load_state_dict keeps
    storage) changes the signature and triggers a fresh capture.
    """

    def __init__(self, model, use_graph: bool = True, warmup: int = 2, threshold: float = 0.5):
        self.model, self.use_graph, self.warmup, self.threshold = model, use_graph, max(1, warmup), threshold
        self._graphs = {}
        self._queue = None

    def _forward(self, x, target):
        with torch.no_grad():
            if not x.is_cuda:
                out = self.model(x)
            else:
                if self._queue is None:
                    self._queue = StepQueue()
                # eval mode: what gets recorded are saved-statistics blocks for a backward pass that never comes
                self._queue.drop = not self.model.training
                with self._queue.active():
                    out = self.model(x)
                if self.model.training:        # the branch stream's running-statistics launches: joined (no backward does it)
                    from .net import join_side_stream
                    join_side_stream(x.device)
            counts = None
            if target is not None:
                from .ops import seg_counts
                counts = seg_counts(out, target, self.threshold)
        return out, counts

    def _signature(self, x, target):
        # (storage addresses: a captured graph holds raw pointers to every parameter and buffer)
        ptrs = hash(tuple(t.data_ptr() for t in list(self.model.parameters()) + list(self.model.buffers())))
        return (tuple(x.shape), x.dtype, None if target is None else tuple(target.shape), self.model.training, ptrs)

    def _capture(self, x, target):
        from .data import GPU_CAPTURE_LOCK
        with GPU_CAPTURE_LOCK:
            static_x = x.clone()
            static_t = target.clone() if target is not None else None
            snap = None
            if self.model.training:            # warm-up forwards must not leave extra running-statistics updates behind
                bufs = list(self.model.buffers())
                snap = [b.detach().clone() for b in bufs]
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(self.warmup):
                    self._forward(static_x, static_t)
                if snap is not None:
                    with torch.no_grad():
                        for b, v in zip(bufs, snap):
                            b.copy_(v)
            torch.cuda.current_stream().wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                out, counts = self._forward(static_x, static_t)
            return graph, static_x, static_t, out, counts

    def __call__(self, x, target=None):
        """-> logits, or (logits, counts) when integer label maps `target` (N,H,W) are given."""
        if not (self.use_graph and x.is_cuda):
            out, counts = self._forward(x, target)
            return out if target is None else (out, counts)
        sig = self._signature(x, target)
        entry = self._graphs.get(sig)
        if entry is None:
            if len(self._graphs) >= 8:         # stale captures (old parameter storage, other shapes) hold memory pools
                self._graphs.clear()
            entry = self._graphs[sig] = self._capture(x, target)
        graph, static_x, static_t, out, counts = entry
        static_x.copy_(x)
        if static_t is not None:
            static_t.copy_(target)
        graph.replay()
        return out if target is None else (out, counts)
