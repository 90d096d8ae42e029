"""Is the graphed train step bound by the GPU or by the host's hipGraphLaunch?  Times (a) the host side of 20
back-to-back replays (until the last replay() call returns), (b) the same with the final synchronize, (c) one replay +
synchronize in isolation, and prints the box's CPU model and GPU clocks beside them."""
import os, subprocess, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "medical-transformer_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import medt_amd
from bench import build_model
from medt_amd.optim import FlatAdam
from medt_amd.trainer import TrainStep

dev = torch.device("cuda:0")
torch.manual_seed(3000)
model = build_model("MedT", 128, dev); model.train()
opt = FlatAdam(list(model.parameters()), lr=1e-3, weight_decay=1e-5)
x = torch.rand(4, 3, 128, 128, device=dev); y = torch.randint(0, 2, (4, 128, 128), device=dev)
ts = TrainStep(model, opt, medt_amd.cross_entropy)
for _ in range(5):
    ts(x, y)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(20):
        ts(x, y)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"20 replays: host-side {1e3*(t1-t0)/20:.3f} ms/step, with sync {1e3*(t2-t0)/20:.3f} ms/step")
lat = []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ts(x, y); t1 = time.perf_counter(); torch.cuda.synchronize()
    lat.append((1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t0)))
print("single replay: host ms / total ms:", [f"{a:.2f}/{b:.2f}" for a, b in lat])
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); e0.record()
for _ in range(20):
    ts(x, y)
e1.record(); torch.cuda.synchronize()
print(f"GPU-event time per step over 20 replays: {e0.elapsed_time(e1)/20:.3f} ms")
try:
    print(subprocess.run("lscpu | grep -E 'Model name|^CPU\\(s\\)|MHz'; rocm-smi --showclocks | grep -E 'sclk|mclk|fclk' | head -4; rocm-smi --showpower | grep -i power | head -2",
                         shell=True, capture_output=True, text=True, timeout=30).stdout)
except Exception as e:
    print("no lscpu/rocm-smi:", e)
