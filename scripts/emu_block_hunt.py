"""Random AxialBlock / _dynamic / _wopos configurations (planes, input planes with / without the downsample path, stride, map size,
batch, BatchNorm groups, mode) through net.axial_block_forward on the CPU lane emulator against float64 autograd through the oracle's
axial_block (pinned to the reference).   python scripts/emu_block_hunt.py <seed> <count>      (no GPU)"""
import ctypes as C
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "medical-transformer_amd"), ROOT]
import torch  # noqa: E402
import helpers as H  # noqa: E402
import lib as droplib  # noqa: E402
import test_lane_emu as T  # noqa: E402
from emu_device import emulated_device  # noqa: E402
from medt_amd import _lib as L, net  # noqa: E402
from oracle import medt_oracle as O  # noqa: E402

lib = C.CDLL(T.build_emulator())
for name, (res, args) in L.SIGNATURES.items():
    fn = getattr(lib, name)
    fn.restype, fn.argtypes = res, args
ax = droplib.models.axialnet
rng = random.Random(int(sys.argv[1]))
n_ok = n_bad = 0
while n_ok + n_bad < int(sys.argv[2]):
    cls = rng.choice([ax.AxialBlock, ax.AxialBlock_dynamic, ax.AxialBlock_wopos])
    planes = rng.choice([16, 32, 64])                       # attention width (base_width 64); the block puts out 2 * planes
    inplanes = rng.choice([2 * planes, 2 * planes, planes, 16, 8])
    stride = rng.choice([1, 1, 2])
    S = rng.choice([4, 8, 16])
    N = rng.choice([1, 2, 4, 8])
    groups = rng.choice([g for g in (1, 2, N) if N % g == 0])
    training = rng.random() < 0.7
    if training and (N // groups) * (S // stride) ** 2 < 4:
        continue
    down = None
    if stride != 1 or inplanes != 2 * planes:
        down = torch.nn.Sequential(ax.conv1x1(inplanes, 2 * planes, stride), torch.nn.BatchNorm2d(2 * planes))
    blk = cls(inplanes, planes, stride, down, groups=8, base_width=64, kernel_size=S)
    st = O.randomize_state({k: v.clone() for k, v in blk.state_dict().items()}, 31)
    blk.load_state_dict(st)
    for p in blk.parameters():
        p.requires_grad_(True)                          # (the reference freezes the gates by default; here they are trained)
    blk.train(training)
    cfg = (cls.__name__, inplanes, planes, stride, S, N, groups, "train" if training else "eval")
    g = torch.Generator().manual_seed(n_ok + n_bad)
    x = torch.randn((N, inplanes, S, S), generator=g).relu_()
    dout = torch.randn((N, 2 * planes, S // stride, S // stride), generator=g)
    try:
        with emulated_device(lib):
            xg = x.clone().requires_grad_(True)
            y = net.axial_block_forward(blk, xg, groups)
            (y * dout).sum().backward()
        ost = O.clone_state({("m." + k): v for k, v in st.items()}, torch.float64, requires_grad=True)
        xo = x.double().requires_grad_(True)
        yo = O.axial_block(xo, ost, "m", stride, training, groups)
        (yo * dout.double()).sum().backward()
        assert H.rel_err(y.detach().double(), yo.detach()) < 1e-3, ("y", H.rel_err(y.detach().double(), yo.detach()))
        assert H.rel_err(xg.grad.double(), xo.grad) < 1e-3, ("dx", H.rel_err(xg.grad.double(), xo.grad))
        gmax = max(v.grad.abs().max().item() for v in ost.values() if v.grad is not None)
        for k, p in blk.named_parameters():
            want = ost["m." + k].grad
            if want is None:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
                continue
            err = (p.grad.double() - want).abs().max().item() / max(want.abs().max().item(), 1e-3 * gmax)
            assert err < 1e-3, (k, err)
        n_ok += 1
    except Exception as e:  # noqa: BLE001
        n_bad += 1
        print("FAIL", cfg, type(e).__name__, str(e)[:300].replace("\n", " "))
print("ok", n_ok, "bad", n_bad)
