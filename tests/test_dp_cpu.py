"""Data-parallel host logic on CPU: world_size 2 over gloo (the GPU path uses the same code over RCCL)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as H  # noqa: F401


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from medt_amd import dp
    torch.manual_seed(100 + rank)                       # deliberately different replicas
    model = torch.nn.Sequential(torch.nn.Conv2d(3, 4, 3), torch.nn.BatchNorm2d(4), torch.nn.Conv2d(4, 2, 1))
    frozen = torch.nn.Parameter(torch.ones(3), requires_grad=False)      # like the gates before epoch 10
    model.register_parameter("gate", frozen)
    dp.broadcast_parameters(model)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    x = torch.full((2, 3, 8, 8), float(rank + 1))
    model(x).sum().backward()
    local = [p.grad.clone() for p in model.parameters() if p.grad is not None]
    bucket = dp.allreduce_gradients(model)
    bucket = dp.allreduce_gradients(model, bucket)      # second call reuses the bucket (averaging an average is a no-op)
    avg = [p.grad.clone() for p in model.parameters() if p.grad is not None]
    # plain numpy payloads: torch tensors would travel through shared-memory handles that die with the worker
    q.put((rank, {k: v.numpy() for k, v in sd.items()}, [g.numpy() for g in local], [g.numpy() for g in avg], bucket.numel))
    dist.destroy_process_group()


def test_broadcast_and_flat_bucket_allreduce():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in procs), key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, sd0, l0, a0, n0), (_, sd1, l1, a1, n1) = res
    import numpy as np
    for k in sd0:
        assert np.array_equal(sd0[k], sd1[k]), k           # replicas identical after the broadcast
    assert n0 == n1 == sum(g.size for g in l0)             # frozen parameter excluded identically on every rank
    for g0, g1, m0, m1 in zip(l0, l1, a0, a1):
        want = (g0 + g1) / 2
        assert np.allclose(m0, want, atol=1e-6) and np.allclose(m1, want, atol=1e-6)
