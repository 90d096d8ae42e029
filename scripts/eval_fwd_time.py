"""Eager eval-mode forward time of MedT (bench.py's fwd_ms_per_image leg on its own), before and after a captured training step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "medical-transformer_amd"), ROOT]
import torch
import lib as droplib
import medt_amd
from medt_amd.optim import FlatAdam
from medt_amd.trainer import TrainStep
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = droplib.models.axialnet.MedT(img_size=128, imgchan=3).to(dev)
x = torch.rand(4, 3, 128, 128, device=dev)
y = torch.randint(0, 2, (4, 128, 128), device=dev)


def evalt(tag):
    model.eval()
    with torch.no_grad():
        for _ in range(3):
            model(x)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(20):
            model(x)
        torch.cuda.synchronize()
        print(tag, "eval fwd ms/image %.3f" % ((time.perf_counter() - t) / 20 / 4 * 1e3), flush=True)
    model.train()


evalt("fresh model:")
opt = FlatAdam(list(model.parameters()), lr=1e-3)
step = TrainStep(model, opt, medt_amd.cross_entropy, use_graph="--eager" not in sys.argv)
for _ in range(8):
    step(x, y)
torch.cuda.synchronize()
evalt("after 8 training steps (%s):" % ("eager" if "--eager" in sys.argv else "hipGraph"))
