import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "medical-transformer_amd"), os.path.join(ROOT, "tests")]
import torch
import helpers as H
import lib as droplib
import medt_amd
from medt_amd.optim import FlatAdam
from medt_amd.trainer import TrainStep
from medt_amd import defer

dev = torch.device("cuda:0")
name, S, N = "MedT", 128, 2
st = H.seeded_state(name, S, 33)
x, y = H.seeded_input(34, N, 3, S)
x, y = x.to(dev), y.to(dev)
res = {}
for tag, use_graph, dfr in (("eager", False, False), ("eager_defer", False, True), ("graph", True, False), ("graph_defer", True, True)):
    defer.ENABLED = dfr
    model = droplib.models.axialnet.MedT(img_size=S, imgchan=3).to(dev)
    model.load_state_dict(st)
    model.train()
    opt = FlatAdam(list(model.parameters()), lr=1e-3, weight_decay=1e-5)
    step = TrainStep(model, opt, medt_amd.cross_entropy, use_graph=use_graph, warmup=2)
    l = [step(x, y).item() for _ in range(2)]
    torch.cuda.synchronize()
    g = opt.groups[0]
    res[tag] = (l, g.flat_g.clone(), [(k, p.numel()) for k, p in model.named_parameters() if id(p) in opt._member])
    print(tag, l)
base = res["eager"]
for tag in ("eager_defer", "graph", "graph_defer"):
    d = (res[tag][1] - base[1]).abs()
    print(tag, "max grad diff", d.max().item(), "of", base[1].abs().max().item())
    off = 0
    bad = []
    for k, n in base[2]:
        e = d[off:off + n].max().item()
        sc = base[1][off:off + n].abs().max().item()
        if e > 1e-3 * max(sc, 1e-6):
            bad.append((k, e, sc))
        off += n
    print("  bad params:", len(bad), bad[:12])
