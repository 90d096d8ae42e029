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


def _run(cmd, env, retries=0):
    r = subprocess.run(cmd, cwd=H.ROOT, env=env, capture_output=True, text=True, timeout=900)
    if r.returncode != 0 and retries:
        # the torchrun worker was seen to abort (SIGABRT) once in ~10 launches on the shared GPU boxes, right after other
        # GPU tests of this process; the first attempt's stderr is kept next to the profiles for inspection
        out = os.path.join(H.ROOT, "gpurun_out")
        if os.path.isdir(out):
            with open(os.path.join(out, "dist_first_attempt_stderr.log"), "w") as f:
                f.write(r.stderr[-20000:])
        return _run(cmd, env, retries - 1)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    return json.loads(line), r.stderr


def test_forced_collectives_over_rccl_match_single_process():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("MEDT_FORCE_DIST", None)
    plain, _ = _run([sys.executable, "bench.py", *ARGS], env)
    env["MEDT_FORCE_DIST"] = "1"
    dist_, err = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                       "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), "bench.py", *ARGS], env, retries=1)
    assert plain["collective"] is None and "nccl" in dist_["collective"]
    assert dist_["collective_in_graph"] is True, dist_["collective"]       # RCCL all-reduce + Adam are graph nodes
    # the collective costs one more graph node, not a host round trip: the forced-RCCL step stays within 10 % of the plain one
    # (measured ~1-2 %; the bound leaves room for box-to-box noise)
    assert dist_["ms_per_step"] <= 1.10 * plain["ms_per_step"], (plain["ms_per_step"], dist_["ms_per_step"])
    assert "process group up, backend=nccl" in err
    assert dist_["n_gpus"] == 1 and dist_["hip_graph"]
    # not bit-equal: the relative-table gradients are accumulated with LDS float atomics (order varies run to run) and
    # training-mode BatchNorm amplifies the last bit over the 6 updates; measured 2e-5, same as two plain runs
    assert abs(dist_["final_loss"] - plain["final_loss"]) <= 5e-4 * abs(plain["final_loss"]), (plain, dist_)
    out = os.path.join(H.ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "dist_forced_rccl.log"), "w") as f:
            f.write(err + "\n" + json.dumps(dist_) + "\nplain run: " + json.dumps(plain) + "\n")
