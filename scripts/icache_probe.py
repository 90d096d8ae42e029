"""Are the fused small-layer kernels (25-100 KB of straight-line code, executed once per launch) bound by COLD
INSTRUCTION FETCH?  Times one AxialAttention_wopos layer's forward+backward kernels (a) back to back (the code stays in
the instruction cache) and (b) with other large kernels of the library in between (as in the training step, where
every variant runs once per step).  Run under `rocprofv3 --kernel-trace --stats`, or read the HIP-event times printed."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "medical-transformer_amd"), ROOT]
import torch  # noqa: E402
import lib as droplib  # noqa: E402

dev = torch.device("cuda:0")
ax = droplib.models.axialnet
torch.manual_seed(0)


def layer(C, L, width):
    m = ax.AxialAttention_wopos(C, C, groups=8, kernel_size=L, stride=1, width=width).to(dev)
    m.train()
    m.bn_groups = 16
    return m


# layer2_p-like: 64 patch images (16 groups of 4), C = 64, 8x8 maps; and four OTHER variants as cache polluters
main = layer(64, 8, False)
x = torch.randn(64, 64, 8, 8, device=dev, requires_grad=True)
others = [(layer(32, 16, True), torch.randn(64, 32, 16, 16, device=dev, requires_grad=True)),
          (layer(32, 16, False), torch.randn(64, 32, 16, 16, device=dev, requires_grad=True)),
          (layer(128, 4, True), torch.randn(64, 128, 4, 4, device=dev, requires_grad=True)),
          (layer(128, 4, False), torch.randn(64, 128, 4, 4, device=dev, requires_grad=True))]


FWD_ONLY = os.environ.get("ICACHE_FWD_ONLY", "1") == "1"      # forward only: the fused kernel + its tiny finalisation


def fb(m, t):
    if FWD_ONLY:
        with torch.no_grad():
            m(t)
        return
    y = m(t)
    y.sum().backward()


def timed(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def capture(fn):
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    return g


def main_only():
    for _ in range(8):
        fb(main, x)


def interleaved():
    for _ in range(2):
        for m, t in others:
            fb(main, x)
            fb(m, t)


def others_only():
    for _ in range(2):
        for m, t in others:
            fb(m, t)


g_a, g_b, g_c = capture(main_only), capture(interleaved), capture(others_only)
ta, tb, tc = timed(g_a.replay, 50), timed(g_b.replay, 50), timed(g_c.replay, 50)
print("forward only" if FWD_ONLY else "forward + backward")
print(f"8 x main layer fwd+bwd back to back: {ta:.1f} us -> {ta / 8:.2f} us per fwd+bwd (warm code)")
print(f"8 x (main + another variant), 8 x another variant alone: {tb:.1f} us, {tc:.1f} us -> main between other variants: "
      f"{(tb - tc) / 8:.2f} us per fwd+bwd")
