"""`medical-transformer_amd/train.py` ITSELF under torch.distributed with world 4 (gloo, CPU) -- round-5 verdict, item 7.

The reference's multi-GPU path is nn.DataParallel(device_ids=[0,1]) inside one process (train.py:104-107); the drop-in CLI runs one
process per GPU.  What only shows up with several processes driving the CLI: DistributedSampler padding on a dataset that is not a
multiple of the global batch (a second, shorter batch shape every epoch), rank 0 validating and writing checkpoints (--save_freq 1)
while the other ranks have already entered the next epoch's first all-reduce, the gates joining the trained set after epoch 10
(reference train.py:169-171) on all ranks at the same step.  tests/cli_dp_driver.py substitutes a CPU toy network, the loss and the
Adam arithmetic (the three things that need the MI355X); everything else is the product's train.py / TrainStep / FlatAdam / dp.

Checked: no hang, every rank exits 0; replicas bit-identical after every epoch; every step's update == Adam on the AVERAGE of the four
shards' gradients (plain autograd in this process, each shard with its own BatchNorm statistics) -- i.e. a single-process run on the
averaged gradients; the gate is untouched through epoch 10 and trained with its own step counter afterwards; rank 0's checkpoint of
every epoch exists and the last one equals its final weights."""
import copy
import os
import subprocess
import sys

import numpy as np
import torch
import torch.nn.functional as F

import helpers as H
from test_dp_cpu import _free_port

WORLD, EPOCHS, NIMG, GBATCH, LR, WD = 4, 12, 10, 8, 1e-3, 1e-5


def test_train_cli_world4_gloo(tmp_path):
    from cli_dp_driver import ToyNet
    data, out = str(tmp_path / "data"), str(tmp_path / "run")
    port = _free_port()
    procs = []
    for r in range(WORLD):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(WORLD),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen(
            [sys.executable, os.path.join(os.path.dirname(__file__), "cli_dp_driver.py"), str(tmp_path / f"rec{r}.pt"),
             "--train_dataset", data, "--val_dataset", data, "--direc", out, "--batch_size", str(GBATCH), "--epochs", str(EPOCHS),
             "--save_freq", "1", "--modelname", "gatedaxialunet", "--learning_rate", str(LR), "--imgsize", "32", "--gray", "no",
             "--synthetic", str(NIMG), "--device", "cpu", "--eager"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("train.py under world 4 hung")
        outs.append((p.returncode, o, e))
    for rc, o, e in outs:
        assert rc == 0, e[-3000:]
    assert "Let's use 4 GPUs!" in outs[0][1] and f"epoch [{EPOCHS - 1}/{EPOCHS}], loss:" in outs[0][1]
    assert all("epoch [" not in o for _, o, _ in outs[1:])                      # rank 0 reports
    recs = [torch.load(str(tmp_path / f"rec{r}.pt"), weights_only=False) for r in range(WORLD)]
    # 10 images over 4 ranks: the sampler pads to 12, 3 per rank, per-rank batch 2 -> batches of 2 and 1 every epoch, on every rank
    per_epoch = 2
    for rec in recs:
        assert len(rec["steps"]) == EPOCHS * per_epoch
        assert [tuple(s["x"].shape) for s in rec["steps"][:2]] == [(2, 3, 32, 32), (1, 3, 32, 32)]
        assert len(rec["epochs"]) == EPOCHS
    # replicas bit-identical after every epoch (rank 0 validates in train mode in between: that moves ITS running statistics only)
    for e in range(EPOCHS):
        for r in range(1, WORLD):
            for k, v in recs[0]["epochs"][e].items():
                assert torch.equal(v, recs[r]["epochs"][e][k]), (e, r, k)
    # the gate: frozen through epoch 10, a second bucket with its own step counter from epoch 11 on
    gate = [recs[0]["epochs"][e]["gate"].item() for e in range(EPOCHS)]
    assert all(g == 0.5 for g in gate[:11]) and gate[11] != 0.5, gate
    n_main = 3 * 4 * 9 + 4 + 4 + 4 * 2 + 2
    assert recs[0]["steps"][0]["groups"] == [n_main] and recs[0]["steps"][-1]["groups"] == [n_main, 1]
    assert all(rec["steps"][-1]["groups"] == [n_main, 1] for rec in recs)
    # every step == Adam on the average of the four shards' gradients at rank 0's pre-step weights
    b1, b2, eps = 0.9, 0.999, 1e-8
    m, v, t = {}, {}, {}
    steps0 = recs[0]["steps"]
    worst = 0.0
    for s, st in enumerate(steps0):
        model = ToyNet()
        model.load_state_dict(st["pre"])
        model.train()
        names = set(st["trainable"]) - {"unused.weight", "unused.bias"}
        grads = {k: 0.0 for k in names}
        for r in range(WORLD):
            rep = copy.deepcopy(model)
            for k, p in rep.named_parameters():
                p.requires_grad_(k in st["trainable"])
            loss = F.cross_entropy(rep(recs[r]["steps"][s]["x"]), recs[r]["steps"][s]["y"])
            assert abs(loss.item() - recs[r]["steps"][s]["loss"]) < 1e-6
            loss.backward()
            for k, p in rep.named_parameters():
                if k in names:
                    grads[k] = grads[k] + p.grad.double() / WORLD
        post = steps0[s + 1]["pre"] if s + 1 < len(steps0) else recs[0]["final"]
        for k in names:
            g = grads[k] + WD * st["pre"][k].double()
            m[k] = b1 * m.get(k, 0.0) + (1 - b1) * g
            v[k] = b2 * v.get(k, 0.0) + (1 - b2) * g * g
            t[k] = t.get(k, 0) + 1
            upd = LR / (1 - b1 ** t[k]) * m[k] / ((v[k] / (1 - b2 ** t[k])).sqrt() + eps)
            err = (post[k].double() - (st["pre"][k].double() - upd)).abs().max().item() / LR
            worst = max(worst, err)
            assert err < 5e-3, (s, k, err)
        for k in ("unused.weight", "unused.bias") + (() if "gate" in names else ("gate",)):
            assert torch.equal(post[k], st["pre"][k]), (s, k)
    assert t["gate"] == per_epoch and t["c1.weight"] == EPOCHS * per_epoch
    print(f"world 4 through train.py: {len(steps0)} steps, worst update deviation {worst:.1e} of one lr step")
    # rank 0's checkpoints: one per epoch (--save_freq 1) + its validation PNGs; the last one holds its final state
    for e in range(EPOCHS):
        assert os.path.exists(os.path.join(out, str(e), "gatedaxialunet.pth"))
        assert len(os.listdir(os.path.join(out, str(e)))) == NIMG + 1
    last = torch.load(os.path.join(out, str(EPOCHS - 1), "gatedaxialunet.pth"))
    for k, w in recs[0]["final"].items():
        assert torch.equal(last[k], w), k
    assert torch.equal(torch.load(out + "final_model.pth")["c1.weight"], recs[0]["final"]["c1.weight"])
