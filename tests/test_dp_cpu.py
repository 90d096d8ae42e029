"""Data-parallel PRODUCT path on CPU: world_size 2 over gloo.

What runs here is the code train.py / bench.py run per step -- medt_amd.trainer.TrainStep (eager branch) driving
medt_amd.optim.FlatAdam: slot adoption, identical exclusion of gradient-less parameters on every rank, ONE
all_reduce(SUM) of each flat bucket, the 1/world factor folded into the Adam update, late-joining gates as a second
group (train.py:169-171).  Only the Adam *kernel* is substituted (medt_adam_step is a HIP kernel; the CPU stand-in
below restates its arithmetic and lives in this test, not in the product).  The expectation is a single-process run
of torch.optim.Adam on the shard-averaged gradients with per-shard BatchNorm statistics -- what the reference's
nn.DataParallel (train.py:104-107) computes.
"""
import copy
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn.functional as F

import helpers as H  # noqa: F401

STEPS, FLIP_AT, LR, WD = 5, 2, 1e-2, 1e-5


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class ToyNet(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.c1 = torch.nn.Conv2d(3, 4, 3, padding=1, bias=False)    # (a bias before BatchNorm has a ~0 gradient: Adam would amplify its rounding)
        self.bn = torch.nn.BatchNorm2d(4)
        self.c2 = torch.nn.Conv2d(4, 2, 1)
        self.unused = torch.nn.Conv2d(4, 4, 1)                                  # never called: MedT's conv1 / adjust_p (Q5)
        self.gate = torch.nn.Parameter(torch.tensor(0.5), requires_grad=False)  # frozen until FLIP_AT, like f_qr...

    def forward(self, x):
        return self.c2(torch.relu(self.bn(self.c1(x)))) * self.gate


def _batch(step, rank=None, world=2):
    g = torch.Generator().manual_seed(7000 + step)
    x = torch.rand(4, 3, 8, 8, generator=g)
    y = torch.randint(0, 2, (4, 8, 8), generator=g)
    if rank is None:
        return x, y
    n = 4 // world
    return x[rank * n:(rank + 1) * n], y[rank * n:(rank + 1) * n]


def _make_cpu_adam():
    from medt_amd.optim import FlatAdam

    class CpuFlatAdam(FlatAdam):
        """FlatAdam with medt_adam_step's arithmetic (csrc/elementwise.hip adam_step) restated in torch for CPU tensors."""

        def _launch_adam(self, g, gscale):
            b1, b2 = self.betas
            g.state[0] += 1
            t = float(g.state[0])
            grad = g.flat_g * gscale + self.weight_decay * g.flat_p
            g.exp_avg.mul_(b1).add_(grad, alpha=1 - b1)
            g.exp_avg_sq.mul_(b2).addcmul_(grad, grad, value=1 - b2)
            denom = (g.exp_avg_sq.sqrt() / (1 - b2 ** t) ** 0.5).add_(self.eps)
            g.flat_p.addcdiv_(g.exp_avg, denom, value=-self.lr / (1 - b1 ** t))

    return CpuFlatAdam


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from medt_amd import dp
    from medt_amd.trainer import TrainStep
    torch.manual_seed(100 + rank)                       # deliberately different replicas before the broadcast
    model = ToyNet()
    dp.broadcast_parameters(model)
    opt = _make_cpu_adam()(list(model.parameters()), lr=LR, weight_decay=WD)
    step = TrainStep(model, opt, F.cross_entropy, use_graph=False)
    losses = []
    for s in range(STEPS):
        if s == FLIP_AT:
            for p in model.parameters():
                p.requires_grad = True                  # train.py:169-171
        x, y = _batch(s, rank, world)
        losses.append(float(step(x, y).detach()))
    sd = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
    q.put((rank, sd, [g.numel for g in opt.groups], losses))
    dist.destroy_process_group()


def _single_process_expectation():
    torch.manual_seed(100)                              # rank 0's initial weights are what the broadcast installs
    model = ToyNet()
    opt = torch.optim.Adam(list(model.parameters()), lr=LR, weight_decay=WD)
    for s in range(STEPS):
        if s == FLIP_AT:
            for p in model.parameters():
                p.requires_grad = True
        grads = []
        for r in range(2):                              # every shard: same weights, its own BatchNorm batch statistics
            rep = model if r == 0 else copy.deepcopy(model)     # replica 0's running stats are the ones kept
            for p in rep.parameters():
                p.grad = None
            x, y = _batch(s, r)
            F.cross_entropy(rep(x), y).backward()
            grads.append([None if p.grad is None else p.grad.clone() for p in rep.parameters()])
        for p, g0, g1 in zip(model.parameters(), *grads):
            p.grad = None if g0 is None else (g0 + g1) / 2
        opt.step()
    return {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}


def test_trainstep_flatadam_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, sd0, groups0, _), (_, sd1, groups1, _) = res
    want = _single_process_expectation()
    n_main = 3 * 4 * 9 + 4 + 4 + 4 * 2 + 2                     # c1, bn, c2: everything that gets a gradient from step 0
    assert groups0 == groups1 == [n_main, 1]                   # `unused` excluded identically; the gate joins as group 2
    for k in sd0:
        if "running" in k or "num_batches" in k:
            continue                                           # BatchNorm statistics stay local to the shard
        assert np.array_equal(sd0[k], sd1[k]), k               # replicas stay bit-identical
        assert np.allclose(sd0[k], want[k], rtol=1e-5, atol=1e-6), (k, np.abs(sd0[k] - want[k]).max())
    assert np.array_equal(sd0["unused.weight"], want["unused.weight"])          # never touched (no weight decay either)
    for k in ("bn.running_mean", "bn.running_var"):
        assert np.allclose(sd0[k], want[k], rtol=1e-5, atol=1e-6), k            # rank 0 == replica 0


def test_flatadam_refuses_cpu_update_loudly():
    """The product optimizer has no CPU path: the Adam update is a HIP kernel."""
    import pytest
    from medt_amd import MedtError
    from medt_amd.optim import FlatAdam
    p = torch.nn.Parameter(torch.ones(3))
    opt = FlatAdam([p])
    opt.zero_grad()
    (p * 2).sum().backward()
    with pytest.raises(MedtError):
        opt.step()


def test_flatadam_slot_window_and_frozen_parameters():
    """Host-side protocol of the flat gradient slots (medt_amd/optim.py): slots are live only between zero_grad() and
    pack_gradients(); a parameter that has been trained cannot be frozen behind the optimizer's back (torch.optim.Adam
    would skip it, the single flat kernel cannot); one that was never trained stays excluded."""
    import pytest
    from medt_amd import MedtError
    from medt_amd import optim as OPT
    a, b, never = (torch.nn.Parameter(torch.ones(3)), torch.nn.Parameter(torch.ones(2)), torch.nn.Parameter(torch.ones(4)))
    never.requires_grad_(False)
    opt = OPT.FlatAdam([a, b, never])
    assert not opt.step_open
    opt.zero_grad()
    assert opt.step_open
    (a.sum() * 2 + b.sum() * 3).backward()
    opt.pack_gradients()                                   # adopts a and b (their first gradients came through autograd)
    assert not opt.step_open and [g.numel for g in opt.groups] == [5]
    sa = OPT.grad_slot(a)
    assert sa is not None and OPT.grad_slot(never) is None
    assert OPT.live(sa) is None                            # window closed: a backward now must go through autograd
    opt.zero_grad()
    assert OPT.live(sa) is sa
    view, direct = OPT.claim(sa)                           # what a slot-aware backward does: write the slot directly ...
    assert direct and view.data_ptr() == opt.groups[0].flat_g.data_ptr()
    view.fill_(7.0)
    tmp, direct2 = OPT.claim(sa)                           # ... and a second use of the parameter accumulates
    assert not direct2
    tmp.fill_(1.0)
    OPT.accumulate(sa, tmp)
    (b.sum() * 3).backward()                               # b through plain autograd in the same step
    opt.pack_gradients()
    assert torch.equal(a.grad, torch.full((3,), 8.0)) and torch.equal(b.grad, torch.full((2,), 3.0))
    assert a.grad.data_ptr() == opt.groups[0].flat_g.data_ptr()
    # freezing a trained parameter
    opt.zero_grad()
    b.requires_grad_(False)
    (a.sum()).backward()
    with pytest.raises(MedtError):
        opt.pack_gradients()


def test_flatadam_backward_outside_the_slot_window_replaces_the_bucket():
    """zero_grad / backward / pack, then `model.zero_grad()` (torch's, not FlatAdam's) + a hand-written
    backward + pack: the gradients of that second backward arrive as fresh `.grad` tensors while the slots still hold the
    first step's gradient under the first step's stamp.  The bucket must then hold G_new, not G_prev + G_new; and a backward
    with NO zero_grad at all must accumulate (torch semantics: `.grad` is the slot, autograd adds in place)."""
    from medt_amd import optim as OPT
    a, b = torch.nn.Parameter(torch.ones(3)), torch.nn.Parameter(torch.ones(2))
    opt = OPT.FlatAdam([a, b])
    opt.zero_grad()
    (a.sum() * 2 + b.sum() * 3).backward()
    opt.pack_gradients()
    flat = opt.groups[0].flat_g
    assert torch.equal(flat, torch.tensor([2.0, 2, 2, 3, 3]))
    for p in (a, b):                                        # what nn.Module.zero_grad() does
        p.grad = None
    (a.sum() * 5 + b.sum() * 7).backward()
    opt.pack_gradients()
    assert torch.equal(flat, torch.tensor([5.0, 5, 5, 7, 7])), flat
    assert a.grad.data_ptr() == flat.data_ptr()
    (a.sum() * 1 + b.sum() * 1).backward()                  # no zero_grad of any kind: accumulation, as in torch
    opt.pack_gradients()
    assert torch.equal(flat, torch.tensor([6.0, 6, 6, 8, 8])), flat
    opt.zero_grad()                                         # and the normal protocol still works afterwards
    (a.sum() * 4 + b.sum() * 4).backward()
    opt.pack_gradients()
    assert torch.equal(flat, torch.tensor([4.0, 4, 4, 4, 4])), flat


# ------------------------------------------------------------------------------------------------------------------------ #
# The same, with a REAL model: lib.models.axialnet.gated (ResAxialAttentionUNet of AxialBlock_dynamic: position tables, gates
# that join the trained set mid-run) at 32 px through the product's own modules,
# autograd Functions, StepQueue, gradient slots, FlatAdam (medt_adam_step) and the C ABI on the emulated device
# (tests/emu_device.py: libmedt_emu.so = every kernel source compiled for the CPU lane emulator), two ranks over gloo.
# ------------------------------------------------------------------------------------------------------------------------ #
R_STEPS, R_FLIP_AT, R_S, R_LR = 2, 1, 32, 1e-3         # (an emulated pass costs ~20 s whatever the image size: two steps, the gates join at the second)


def _emu_lib():
    import ctypes
    import test_lane_emu as T
    from medt_amd import _lib as L
    lib = ctypes.CDLL(T.build_emulator())
    for name, (res, args) in L.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    return lib


def _real_batch(step, rank):
    from emu_device import DeviceTensor
    g = torch.Generator().manual_seed(8000 + 10 * step + rank)
    x = torch.rand(2, 3, R_S, R_S, generator=g)
    y = torch.randint(0, 2, (2, R_S, R_S), generator=g)
    return x.as_subclass(DeviceTensor), y.as_subclass(DeviceTensor)


def _real_model(seed):
    import lib as droplib
    torch.manual_seed(seed)
    return droplib.models.axialnet.gated(img_size=R_S, imgchan=3).train()


def _real_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import medt_amd
    from emu_device import emulated_device
    from medt_amd import dp
    from medt_amd.optim import FlatAdam
    from medt_amd.trainer import TrainStep
    with emulated_device(_emu_lib()):
        model = _real_model(100 + rank)                 # deliberately different replicas before the broadcast
        dp.broadcast_parameters(model)
        opt = FlatAdam(list(model.parameters()), lr=R_LR, weight_decay=WD)
        step = TrainStep(model, opt, medt_amd.cross_entropy, use_graph=False)
        trace = []
        for s in range(R_STEPS):
            if s == R_FLIP_AT:
                for p in model.parameters():
                    p.requires_grad = True              # train.py:169-171: the gates join as a second group
            pre = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
            x, y = _real_batch(s, rank)
            loss = float(step(x, y).detach())
            grads = {k: p.grad.detach().numpy().copy() for k, p in model.named_parameters() if p.grad is not None}
            trace.append((pre, grads, loss))            # grads: the all-reduced SUM (1/world is folded into the Adam kernel)
        sd = {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}
        q.put((rank, sd, [g.numel for g in opt.groups], trace if rank == 0 else [t[2] for t in trace]))
    dist.destroy_process_group()


def _real_shard_job(args):
    """One shard's gradients at the weights rank 0 had BEFORE `step`: the same emulated kernels through plain autograd (no gradient
    slots, no recorded jobs, no bucket), the shard's own BatchNorm statistics.  (Its own process: an emulated pass takes ~20 s.)"""
    pre, step, r = args
    torch.set_num_threads(1)
    import medt_amd
    from emu_device import emulated_device
    with emulated_device(_emu_lib()):
        model = _real_model(0)
        model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in pre.items()})
        for p in model.parameters():
            p.requires_grad = True
        x, y = _real_batch(step, r)
        loss = medt_amd.cross_entropy(model(x), y)
        loss.backward()
        return {k: p.grad.detach().numpy().copy() for k, p in model.named_parameters() if p.grad is not None}, float(loss.detach())


def test_real_model_data_parallel_world2_on_the_emulated_device():
    """Whole-network training in batch-statistics mode amplifies rounding by ~1e4 (parity floor), and Adam then turns a
    noisy small gradient into a full lr step: comparing WEIGHTS after several steps against a separately run expectation tests
    the conditioning of the network, not the data-parallel code.  So every step is checked on its own, at rank 0's weights:
    (1) the all-reduced bucket == the sum of the two shards' gradients computed by a single process with plain autograd;
    (2) the update == Adam on bucket / world (moments tracked here from the buckets), the late-joining gates starting at t = 1;
    (3) the replicas stay bit-identical; BatchNorm running statistics stay local to the shard."""
    import test_lane_emu as T
    T.build_emulator()                                  # once, before the ranks race for it
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_real_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=900) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, sd0, groups0, trace), (_, sd1, groups1, losses1) = res
    gate_names = ("f_qr", "f_kr", "f_sve", "f_sv")
    n_gates = sum(v.size for k, v in sd0.items() if k.split(".")[-1] in gate_names)
    n_all = sum(v.size for k, v in sd0.items() if "running" not in k and "num_batches" not in k and "flatten_index" not in k)
    assert n_gates == 4 * 16
    # two groups on both ranks: everything that had a gradient from step 0, then the gates (4 per attention layer).  (Tensors
    # that never get a gradient -- MedT's conv1 / adjust_p, SURVEY Q5 -- exist only in the 128-px networks: ToyNet.unused above.)
    assert groups0 == groups1 == [n_all - n_gates, n_gates], (groups0, n_all, n_gates)
    for k in sd0:
        if "running" in k:
            assert not np.array_equal(sd0[k], sd1[k]) or np.all(sd0[k] == sd0[k].flat[0]), k     # per-shard statistics
        elif "num_batches" not in k and "flatten_index" not in k:
            assert np.array_equal(sd0[k], sd1[k]), k                                              # replicas bit-identical
    b1, b2, eps = 0.9, 0.999, 1e-8
    m, v, t = {}, {}, {}
    worst_g = worst_p = 0.0
    with ctx.Pool(2 * len(trace)) as pool:              # (step, shard) jobs side by side
        shard = pool.map(_real_shard_job, [(pre, s, r) for s, (pre, _, _) in enumerate(trace) for r in range(2)])
    if True:
        for s, (pre, grads, loss0) in enumerate(trace):
            (g0, l0), (g1, l1) = shard[2 * s], shard[2 * s + 1]
            want = {k: g0[k] + g1[k] for k in g0}
            assert abs(l0 - loss0) < 1e-5 * abs(l0) and abs(l1 - losses1[s]) < 1e-5 * abs(l1)
            trained = {k for k in want if s >= R_FLIP_AT or k.split(".")[-1] not in gate_names}
            assert set(grads) == trained, set(grads) ^ trained
            gmax = max(np.abs(want[k]).max() for k in trained)
            post = trace[s + 1][0] if s + 1 < len(trace) else sd0
            for k in sorted(trained):
                # (1) bucket == sum of the shard gradients (other summation order in the recorded weight-gradient jobs: rounding)
                err = np.abs(grads[k] - want[k]).max() / max(np.abs(want[k]).max(), 1e-3 * gmax)
                worst_g = max(worst_g, err)
                assert err < 1e-4, (s, k, err)
                # (2) the Adam update of medt_adam_step on bucket / world
                g = grads[k].astype(np.float64) / 2 + WD * pre[k].astype(np.float64)
                m[k] = b1 * m.get(k, 0.0) + (1 - b1) * g
                v[k] = b2 * v.get(k, 0.0) + (1 - b2) * g * g
                t[k] = t.get(k, 0) + 1
                upd = R_LR / (1 - b1 ** t[k]) * m[k] / (np.sqrt(v[k] / (1 - b2 ** t[k])) + eps)
                perr = np.abs(post[k] - (pre[k] - upd)).max() / R_LR
                worst_p = max(worst_p, perr)
                assert perr < 2e-3, (s, k, perr)
            for k in pre:                               # everything else is untouched by the step
                if k not in trained and "running" not in k and "num_batches" not in k:
                    assert np.array_equal(pre[k], post[k]), (s, k)
    assert t["layer1.0.hight_block.f_qr"] == R_STEPS - R_FLIP_AT and t["conv1.weight"] == R_STEPS
    print("worst bucket-vs-shard-sum gradient error %.2e (tensor scale); worst update deviation %.2e of one lr step" % (worst_g, worst_p))


# ------------------------------------------------------------------------------------------------------------------------ #
# Two gradient buckets (optim.TWO_BUCKETS): a two-branch network whose backward functions follow the gradient-slot protocol of
# medt_amd.ops (claim the slot, write the parameter gradient directly, call branch_done() where the first branch ends), two
# ranks over gloo: the early all-reduce of segment 0 + the late one of segment 1 == one all-reduce of everything.
# ------------------------------------------------------------------------------------------------------------------------ #
class _SlotLinearFn(torch.autograd.Function):
    """y = x @ w.T with the weight gradient written straight into FlatAdam's slot when one is live (what ConvBlockFn does)."""

    @staticmethod
    def forward(ctx, x, w, last_of_branch):
        from medt_amd import optim as OPT
        ctx.save_for_backward(x, w)
        ctx.slot, ctx.last = OPT.grad_slot(w), last_of_branch
        return x @ w.t()

    @staticmethod
    def backward(ctx, dy):
        from medt_amd import optim as OPT
        x, w = ctx.saved_tensors
        dw, ret = dy.t() @ x, None
        slot = OPT.live(ctx.slot)
        if slot is not None:
            dst, direct = OPT.claim(slot)
            dst.copy_(dw) if direct else OPT.accumulate(slot, dw)
        else:
            ret = dw
        if ctx.last:
            OPT.branch_done(0)
        return dy @ w, ret, None


class _TwoBranch(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a1, self.a2 = torch.nn.Linear(6, 6, bias=False), torch.nn.Linear(6, 3, bias=False)          # "global" branch
        self.b1_p, self.b2_p = torch.nn.Linear(6, 6, bias=False), torch.nn.Linear(6, 3, bias=False)      # "local" branch (*_p)
        self.trunk = torch.nn.Linear(3, 2, bias=False)

    def forward(self, x):
        xa = x.clone().requires_grad_(True)              # (so that a1's backward -- the last node of its branch -- runs)
        ga = _SlotLinearFn.apply(torch.tanh(_SlotLinearFn.apply(xa, self.a1.weight, True)), self.a2.weight, False)
        gb = _SlotLinearFn.apply(torch.tanh(_SlotLinearFn.apply(x.clone().requires_grad_(True), self.b1_p.weight, False)), self.b2_p.weight, False)
        return _SlotLinearFn.apply(ga + gb, self.trunk.weight, False)


def _two_bucket_worker(rank, world, port, q, two):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from medt_amd import dp, optim as OPT
    from medt_amd.trainer import TrainStep
    OPT.TWO_BUCKETS = two
    calls = []
    real = dist.all_reduce
    dist.all_reduce = lambda t, op=None, **k: (calls.append(t.numel()), real(t, op=op, **k))[1]
    torch.manual_seed(5)
    model = _TwoBranch()
    dp.broadcast_parameters(model)
    opt = _make_cpu_adam()(list(model.parameters()), lr=LR, weight_decay=WD)
    step = TrainStep(model, opt, F.cross_entropy, use_graph=False)
    per_step = []
    for s in range(3):
        g = torch.Generator().manual_seed(900 + 10 * s + rank)
        x, y = torch.randn(8, 6, generator=g), torch.randint(0, 2, (8,), generator=g)
        calls.clear()
        step(x, y)
        per_step.append(list(calls))
    q.put((rank, {k: v.detach().numpy().copy() for k, v in model.state_dict().items()}, per_step,
           [tuple(g.bounds) for g in opt.groups]))
    dist.destroy_process_group()


def _run_two_bucket(two):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_two_bucket_worker, args=(r, 2, port, q, two)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=180) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return res


def test_two_gradient_buckets_early_allreduce_world2():
    one, two = _run_two_bucket(False), _run_two_bucket(True)
    n_a, n_b, n_t = 36 + 18, 36 + 18, 6
    # one bucket: a single collective of everything per step
    assert one[0][2] == one[1][2] == [[n_a + n_b + n_t]] * 3 and one[0][3] == [((0, n_a + n_b + n_t),)]
    # two buckets: layout [global + trunk | local]; step 0 adopts (gradients through autograd: one collective); from step 1 on
    # the global segment goes out first (branch_done in a1's backward), the local one after the backward
    assert two[0][3] == two[1][3] == [((0, n_a + n_t), (n_a + n_t, n_a + n_b + n_t))]
    assert two[0][2] == two[1][2] == [[n_a + n_b + n_t], [n_a + n_t, n_b], [n_a + n_t, n_b]], two[0][2]
    for k in one[0][1]:
        assert np.array_equal(two[0][1][k], two[1][1][k]), k                     # replicas bit-identical
        assert np.array_equal(two[0][1][k], one[0][1][k]), k                     # == the single-bucket run, bit for bit
