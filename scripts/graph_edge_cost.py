"""What does a hipGraph pay per dependent kernel, and per cross-stream edge (fork / join)?

Captured with torch.cuda.graph, replayed 200 x, HIP-event timed.  Shapes:
  chain    N tiny kernels on one stream
  par2     two streams x N/2 kernels, ONE fork + ONE join
  pp_k     one stream's chain with k fork/join round trips to a second stream spread along it (each: fork, 1 kernel on the
           side stream, join) -- the same N kernels in total
  fan_k    k side kernels forked off a main chain at k points, joined only at the END (what flush streams do)
  chainL / par2L / 2graphs   (argv[2] = kernel duration in us, default 8) chain and two-branch graph of latency-bound spin kernels,
           and the two branches as two single-stream graphs replayed on two streams with a fork + join per iteration
Prints us per replay and the derived per-kernel / per-edge costs."""
import sys
import torch

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
bufs = [torch.zeros(256, device=dev) for _ in range(4)]


def tiny(i=0):
    bufs[i].add_(1.0)


def timed(build):
    side = torch.cuda.Stream()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        build(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            build(side)
        for _ in range(20):
            g.replay()
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(200):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / 200)
    return best


def chain(side):
    for _ in range(N):
        tiny(0)


def par2(side):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        for _ in range(N // 2):
            tiny(1)
    for _ in range(N // 2):
        tiny(0)
    main.wait_stream(side)


def pp(k):
    def build(side):
        main = torch.cuda.current_stream()
        per = (N - k) // (k + 1)
        done = 0
        for j in range(k):
            for _ in range(per):
                tiny(0)
            done += per
            side.wait_stream(main)
            with torch.cuda.stream(side):
                tiny(0)
            main.wait_stream(side)
            done += 1
        for _ in range(N - done):
            tiny(0)
    return build


def fan(k):
    def build(side):
        main = torch.cuda.current_stream()
        per = (N - k) // (k + 1)
        done = 0
        for j in range(k):
            for _ in range(per):
                tiny(0)
            done += per
            ev = torch.cuda.Event()
            ev.record(main)
            side.wait_event(ev)
            with torch.cuda.stream(side):
                tiny(2)
            done += 1
        for _ in range(N - done):
            tiny(0)
        main.wait_stream(side)
    return build


t_chain = timed(chain)
print(f"chain  N={N}: {t_chain:8.1f} us/replay = {t_chain / N:.2f} us per dependent kernel")
t = timed(par2)
print(f"par2   N={N}: {t:8.1f} us/replay (ideal {t_chain / 2:.1f}): fork+join costs {t - t_chain / 2:.1f} us")
for k in (1, 4, 16):
    t = timed(pp(k))
    print(f"pp_{k:<3d} N={N}: {t:8.1f} us/replay: {(t - t_chain) / k:.1f} us per fork/join round trip")
for k in (1, 4, 16):
    t = timed(fan(k))
    print(f"fan_{k:<2d} N={N}: {t:8.1f} us/replay: {(t - t_chain) / k:+.1f} us per fork (ideal: -{t_chain / N:.1f}, one kernel leaves the chain)")

# ---- the same with latency-bound kernels of a few microseconds (scripts/ubench/spin.hip: 64 workgroups spinning on the wall
# clock -- the step's typical launch), and two single-stream graphs side by side -------------------------------------------------
import ctypes, os
_spin = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "ubench", "libspin.so"))
_spin.spin_launch.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_int, ctypes.c_void_p]
US = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
sink = torch.zeros(4096, device=dev, dtype=torch.int32)


def work(i):
    rc = _spin.spin_launch(torch.cuda.current_stream().cuda_stream, US, 64, sink.data_ptr() + 4096 * i)
    assert rc == 0


def chainL(side):
    for _ in range(N):
        work(0)


def par2L(side):
    main = torch.cuda.current_stream()
    side.wait_stream(main)
    with torch.cuda.stream(side):
        for _ in range(N // 2):
            work(1)
    for _ in range(N // 2):
        work(0)
    main.wait_stream(side)


def halfL(i):
    def build(side):
        for _ in range(N // 2):
            work(i)
    return build


tL = timed(chainL)
print(f"chainL N={N} x {US} us: {tL:8.1f} us/replay = {tL / N:.2f} us per kernel+boundary")
t = timed(par2L)
print(f"par2L  N={N}: {t:8.1f} us/replay (ideal {tL / 2:.1f}): one graph with two branches costs {(t - tL / 2) / (N / 2):+.2f} us per kernel level")


def two_graphs():
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    gs = []
    for i, s in enumerate((s1, s2)):
        with torch.cuda.stream(s):
            halfL(i)(None)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):
                halfL(i)(None)
            gs.append(g)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record(s1)
        for _ in range(200):
            # fork s1 -> s2, both replay, join: what a step made of per-branch graphs does
            s2.wait_stream(s1)
            with torch.cuda.stream(s1):
                gs[0].replay()
            with torch.cuda.stream(s2):
                gs[1].replay()
            s1.wait_stream(s2)
        e1.record(s1)
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / 200)
    return best


t = two_graphs()
print(f"2graphs N={N}: {t:8.1f} us/replay (ideal {tL / 2:.1f}): two single-stream graphs on two streams, fork + join per iteration: {(t - tL / 2):+.1f} us")
