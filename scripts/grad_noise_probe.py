"""Probe: product (fp32, GPU) gradient error vs the fp64 oracle, same metric as the reference's own
fp32-vs-fp64 discrepancy (see DESIGN.md "parity floor")."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "medical-transformer_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
import helpers as H
from oracle import medt_oracle as O
import lib as droplib

dev = torch.device("cuda:0")
for name, S, N, seed in [("gatedaxialunet", 128, 4, 101), ("MedT", 128, 4, 102)]:
    f = {"gatedaxialunet": droplib.models.axialnet.gated, "MedT": droplib.models.axialnet.MedT}[name]
    model = f(img_size=S, imgchan=3).to(dev)
    st = H.seeded_state(name, S, seed)
    model.load_state_dict(st)
    for p in model.parameters():
        p.requires_grad_(True)
    model.train()
    x, y = H.seeded_input(seed + 1, N, 3, S)
    out = model(x.to(dev))
    torch.nn.functional.cross_entropy(out, y.to(dev)).backward()
    ost = O.clone_state(st, torch.float64, requires_grad=True)
    oout = O.forward(name, x.double(), ost, True)
    O.log_nll_loss(oout, y).backward()
    gmax = max(v.grad.norm().item() for v in ost.values() if v.grad is not None)
    rows = []
    for k, p in model.named_parameters():
        g = ost[k].grad
        if g is None or p.grad is None:
            continue
        rows.append(((p.grad.double().cpu() - g).norm().item() / max(g.norm().item(), 1e-3 * gmax), k))
    rows.sort()
    print(name, N, "logits %.1e" % H.rel_err(out, oout), "grad worst %.1e %s median %.1e" % (rows[-1][0], rows[-1][1], rows[len(rows) // 2][0]))
    for r in rows[-5:]:
        print("    %.2e %s" % r)
