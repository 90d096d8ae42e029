"""TEST INFRASTRUCTURE (tests/test_cli_dp_cpu.py): one rank of `medical-transformer_amd/train.py` under torch.distributed (gloo, CPU).

What is tested is the CLI's own multi-process logic -- DistributedSampler on a dataset that is not a multiple of the global
batch (a shorter last batch, the same on every rank), rank-0-only validation + checkpoints while the other ranks run ahead into
the next epoch's collective, the gates joining the trained set after epoch 10 (train.py:169-171 of the reference), FlatAdam's
buckets and the all-reduce -- so the three things that need the MI355X are substituted HERE, in the test, never in the product:
the network (a 3-layer CPU toy with a BatchNorm, a frozen gate and a never-used tensor, installed under the factory name
`gatedaxialunet`), the loss kernel (F.cross_entropy = what medt_ce_fwd/bwd compute) and the Adam kernel (medt_adam_step's
arithmetic restated in torch, as in tests/test_dp_cpu.py).  The real network through the same DP path on the emulated device is
tests/test_dp_cpu.py::test_real_model_data_parallel_world2_on_the_emulated_device.

Every rank records, per training step: the shard it was given, and (rank 0) the weights before the step; per epoch: a copy of its
parameters.  argv: <record.pt> <train.py args...>"""
import os
import runpy
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "medical-transformer_amd")
for p in (ROOT, PKG, os.path.dirname(os.path.abspath(__file__))):
    if p not in sys.path:
        sys.path.insert(0, p)


class ToyNet(torch.nn.Module):
    def __init__(self, img_size=32, imgchan=3):
        super().__init__()
        self.c1 = torch.nn.Conv2d(imgchan, 4, 3, padding=1, bias=False)
        self.bn = torch.nn.BatchNorm2d(4)
        self.c2 = torch.nn.Conv2d(4, 2, 1)
        self.unused = torch.nn.Conv2d(4, 4, 1)                                  # never called: MedT's conv1 / adjust_p
        self.gate = torch.nn.Parameter(torch.tensor(0.5), requires_grad=False)  # frozen until epoch 10, like f_qr ...

    def forward(self, x):
        return self.c2(torch.relu(self.bn(self.c1(x)))) * self.gate


def main():
    record_path, argv = sys.argv[1], sys.argv[2:]
    torch.set_num_threads(1)
    import lib
    import metrics
    from medt_amd import optim, trainer

    torch.manual_seed(100 + int(os.environ.get("RANK", "0")))      # deliberately different replicas before the broadcast
    lib.models.axialnet.gated = ToyNet

    class CpuLoss(torch.nn.Module):
        def forward(self, y_pred, y_true):
            return F.cross_entropy(y_pred, y_true)
    metrics.LogNLLLoss = CpuLoss

    def launch_adam(self, g, gscale):                               # csrc/elementwise.hip adam_step, restated (test_dp_cpu.py)
        b1, b2 = self.betas
        g.state[0] += 1
        t = float(g.state[0])
        grad = g.flat_g * gscale + self.weight_decay * g.flat_p
        g.exp_avg.mul_(b1).add_(grad, alpha=1 - b1)
        g.exp_avg_sq.mul_(b2).addcmul_(grad, grad, value=1 - b2)
        denom = (g.exp_avg_sq.sqrt() / (1 - b2 ** t) ** 0.5).add_(self.eps)
        g.flat_p.addcdiv_(g.exp_avg, denom, value=-self.lr / (1 - b1 ** t))
    optim.FlatAdam._launch_adam = launch_adam

    rec = {"steps": [], "epochs": [], "rank": int(os.environ.get("RANK", "0"))}
    call = trainer.TrainStep.__call__

    def recording_call(self, x, y):
        pre = {k: v.detach().clone() for k, v in self.model.state_dict().items()} if rec["rank"] == 0 else None
        req = [k for k, p in self.model.named_parameters() if p.requires_grad]
        loss = call(self, x, y)
        rec["steps"].append({"x": x.detach().clone(), "y": y.detach().clone(), "pre": pre, "trainable": req,
                             "loss": float(loss.detach()), "groups": [g.numel for g in self.opt.groups]})
        rec["model"] = self.model
        return loss
    trainer.TrainStep.__call__ = recording_call

    # every rank snapshots its parameters when an epoch's last step is behind it: DistributedSampler.set_epoch marks the start of the next
    from torch.utils.data.distributed import DistributedSampler
    set_epoch = DistributedSampler.set_epoch

    def marking_set_epoch(self, epoch):
        if "model" in rec:
            rec["epochs"].append({k: v.detach().clone() for k, v in rec["model"].named_parameters()})
        rec["epoch_starts"] = rec.get("epoch_starts", []) + [len(rec["steps"])]
        return set_epoch(self, epoch)
    DistributedSampler.set_epoch = marking_set_epoch

    sys.argv = [os.path.join(PKG, "train.py")] + argv
    try:
        runpy.run_path(sys.argv[0], run_name="__main__")
    finally:
        if "model" in rec:
            rec["epochs"].append({k: v.detach().clone() for k, v in rec["model"].named_parameters()})
            rec["final"] = {k: v.detach().clone() for k, v in rec["model"].state_dict().items()}
            del rec["model"]
        torch.save(rec, record_path)


if __name__ == "__main__":
    main()
