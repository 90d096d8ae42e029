"""The data-parallel step on hardware: bench.py under torch.distributed.run with one rank and MEDT_FORCE_DIST=1, so the
nccl (== RCCL) process-group init, the flat-bucket all-reduce and the Adam launch captured INSIDE the hipGraph, the
barrier + MAX-over-ranks timing and the 1/world scaling all execute on the MI355X.  With one rank the sum over ranks is
the identity, so the trajectory must equal the plain single-process run."""
import json
import os
import socket
import subprocess
import sys

import pytest

import helpers as H

pytestmark = pytest.mark.gpu
ARGS = ["--gpus", "1", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--no-roofline"]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(cmd, env):
    r = subprocess.run(cmd, cwd=H.ROOT, env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0:                # keep the evidence next to the profiles (no retry: a failure here is a finding)
        out = os.path.join(H.ROOT, "gpurun_out")
        if os.path.isdir(out):
            with open(os.path.join(out, "dist_failure_stderr.log"), "w") as f:
                f.write(f"rc={r.returncode}\n" + r.stderr[-40000:])
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line), r.stderr


def test_forced_collectives_over_rccl_match_single_process():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("MEDT_FORCE_DIST", None)
    plain, _ = _run([sys.executable, "bench.py", *ARGS], env)
    env["MEDT_FORCE_DIST"] = "1"
    dist_, err = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                       "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "bench.py", *ARGS], env)
    assert plain["collective"] is None and "nccl" in dist_["collective"]
    assert dist_["collective_in_graph"] is True, dist_["collective"]       # RCCL all-reduce + Adam are graph nodes
    # round 5: two buckets -- [global branch + trunk | local branch] -- the first all-reduced where the global backward ends
    if os.environ.get("MEDT_TWO_BUCKETS", "1") != "0":
        assert len(dist_["gradient_buckets"][0]) == 2 and sum(dist_["gradient_buckets"][0]) == 1524546, dist_["gradient_buckets"]
    # the collective costs one more graph node, not a host round trip: the forced-RCCL step stays within 4 % of the plain one
    # (measured 1.9 % in round 3; both are medians of five 20-step windows on the same box)
    assert dist_["ms_per_step"] <= 1.04 * plain["ms_per_step"], (plain["ms_per_step"], dist_["ms_per_step"])
    assert "process group up, backend=nccl" in err
    assert dist_["n_gpus"] == 1 and dist_["hip_graph"]
    # With one rank the all-reduce is the identity and 1/world == 1: the 105 updates (5 warm-up + 5 x 20) must be BIT-equal to
    # the plain run's.  (MedT's position-encoded layers all take the single-sweep backward, which has no float atomics; the
    # fused small-layer kernels and the grouped weight-gradient launches reduce in fixed order.)
    assert dist_["final_loss"] == plain["final_loss"], (plain["final_loss"], dist_["final_loss"])
    out = os.path.join(H.ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "dist_forced_rccl.log"), "w") as f:
            f.write(err + "\n" + json.dumps(dist_) + "\nplain run: " + json.dumps(plain) + "\n")


_REFUSAL_SCRIPT = r'''
import os, sys, torch
import torch.distributed as dist
root = os.environ["MEDT_ROOT"]
sys.path[:0] = [os.path.join(root, "medical-transformer_amd"), root, os.path.join(root, "tests")]
import helpers as H
import lib as droplib
import medt_amd
from medt_amd import optim as OPT
from medt_amd.optim import FlatAdam
from medt_amd.trainer import TrainStep

dev = torch.device("cuda:0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % os.environ["MEDT_PORT"], rank=0, world_size=1, device_id=dev)
OPT.FORCE_COLLECTIVES = True                 # one rank, but the collectives run (what MEDT_FORCE_DIST=1 sets at import)
assert OPT.collectives_needed()
name, S, N = "MedT", 128, 2
st = H.seeded_state(name, S, 33)
x, y = H.seeded_input(34, N, 3, S)
x, y = x.to(dev), y.to(dev)
real_all_reduce = dist.all_reduce
calls = {"refused": 0, "eager": 0}

def refusing_all_reduce(t, *a, **k):
    """A process group whose collectives cannot be captured: raises while the stream is capturing, works otherwise."""
    if torch.cuda.is_current_stream_capturing():
        calls["refused"] += 1
        raise RuntimeError("collective refuses stream capture (test double)")
    calls["eager"] += 1
    return real_all_reduce(t, *a, **k)

results = []
for mode in ("graph_refused", "graph_in", "eager"):
    model = droplib.models.axialnet.MedT(img_size=S, imgchan=3).to(dev)
    model.load_state_dict(st)
    model.train()
    opt = FlatAdam(list(model.parameters()), lr=1e-3, weight_decay=1e-5)
    step = TrainStep(model, opt, medt_amd.cross_entropy, use_graph=(mode != "eager"), warmup=2)
    dist.all_reduce = refusing_all_reduce if mode == "graph_refused" else real_all_reduce
    calls["refused"] = calls["eager"] = 0
    try:
        if mode == "eager":
            # The first step a FlatAdam ever sees is its adoption step: the gradients arrive through autograd's `.grad` and the
            # weight-gradient kernels launch immediately (other chunking = other fp32 summation order than the recorded, grouped
            # launches of every later step).  The captured modes spend that step in their rolled-back warm-up; do the same here,
            # so that the three timed steps of all modes run the same kernels.
            snap = step._snapshot()
            step._eager(x, y)
            step._restore(snap)
            torch.cuda.synchronize()
            calls["eager"] = 0
        losses = [step(x, y).item() for _ in range(3)]
    finally:
        dist.all_reduce = real_all_reduce
    torch.cuda.synchronize()
    if mode == "graph_refused":
        assert calls["refused"] == 1, calls                    # the first capture attempt hit the refusal ...
        assert step.collective_in_graph is False               # ... the second left the collective outside the graph
        # warm-up: the adoption step (one collective) + one step with two buckets (global segment early, local segment late);
        # then one per replayed step, behind the replay (no early all-reduce inside a graph that must not hold collectives)
        assert calls["eager"] == (3 if OPT.TWO_BUCKETS else 2) + 3, calls
    if mode == "graph_in":
        assert step.collective_in_graph is True
    g = opt.groups[0]
    results.append((mode, losses, g.flat_p.clone(), g.exp_avg.clone(), g.exp_avg_sq.clone(), g.state.clone(),
                    {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}))
    step._graphs.clear()
    del step
ref = results[-1]                                              # the uncaptured step
for mode, losses, p, m, v, s, bufs in results:
    print(mode, losses, "max |dw| vs eager", (p - ref[2]).abs().max().item(), flush=True)
for mode, losses, p, m, v, s, bufs in results[:-1]:
    assert losses == ref[1], (mode, losses, ref[1])
    assert torch.equal(s, ref[5]), (mode, s, ref[5])           # Adam's step counter and bias corrections: 3 updates, not 3 + warm-up
    assert float(s[0]) == 3.0
    for a, b, what in ((p, ref[2], "weights"), (m, ref[3], "exp_avg"), (v, ref[4], "exp_avg_sq")):
        assert torch.equal(a, b), (mode, what, (a - b).abs().max().item())
    for k in bufs:
        assert torch.equal(bufs[k], ref[6][k]), (mode, k)
torch.cuda.synchronize()
dist.destroy_process_group()
print("refusal ok")
'''


def test_collective_that_refuses_capture_falls_back_outside_the_graph():
    """trainer.TrainStep._capture_locked's `attempt` loop: a process group whose all-reduce raises under stream capture makes
    the first capture fail; the step is then captured WITHOUT the collective and the all-reduce + Adam run behind every
    replay.  Three steps of that, of the normal in-graph capture and of the uncaptured eager step must give bit-identical
    losses, weights, Adam moments / step counters and BatchNorm running statistics (RCCL process group with one rank, in a
    subprocess so that RCCL never lives in the pytest process)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MEDT_ROOT=H.ROOT, MEDT_PORT=str(_free_port()))
    env.pop("MEDT_FORCE_DIST", None)
    r = subprocess.run([sys.executable, "-c", _REFUSAL_SCRIPT], cwd=H.ROOT, env=env, capture_output=True, text=True, timeout=900)
    out = os.path.join(H.ROOT, "gpurun_out")
    if r.returncode != 0 and os.path.isdir(out):
        with open(os.path.join(out, "refusal_failure.log"), "w") as f:
            f.write(f"rc={r.returncode}\n--- stdout\n{r.stdout[-20000:]}\n--- stderr\n{r.stderr[-40000:]}")
    assert r.returncode == 0 and "refusal ok" in r.stdout, (r.returncode, r.stdout[-1500:], r.stderr[-3000:])
