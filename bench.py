#!/usr/bin/env python
"""bench.py -- training images/s of MedT (3x128x128) on N MI355X, plus the attention-kernel roofline
and the CPU baseline.  Contract: see the task description / DESIGN.md "Measurement".

  python bench.py --gpus 1 --steps 20 --warmup 5
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = forward + cross-entropy + backward + (N>1: gradient all-reduce) + Adam on one synthetic batch
already resident in HBM (reference train.py:140,156-161).  Weak scaling: 4 images per GPU.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "medical-transformer_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec (guides/MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)
VALU_PEAK_TFLOPS = 157.3        # fp32 vector peak: 256 CUs x 128 lanes x 2 flop x 2.4 GHz (same guide)
PER_GPU_BATCH = 4               # BASELINE.json configs[2]/[3]: MedT imgsize=128 bs=4 per GPU
IMG = 128


def build_model(name, img, device):
    import lib as droplib
    f = {"MedT": droplib.models.axialnet.MedT, "gatedaxialunet": droplib.models.axialnet.gated, "gated": droplib.models.axialnet.gated,
         "axialunet": droplib.models.axialunet, "logo": droplib.models.axialnet.logo}[name]
    return f(img_size=img, imgchan=3).to(device)


# --------------------------------------------------------------------------- #
# roofline leg: the fused attention kernel on a shape whose traffic exceeds the 256 MB Infinity Cache
# --------------------------------------------------------------------------- #
def roofline_leg(device, C=16, L=64, images=256, iters=20):
    """layer-1 geometry (C=16, G=8, L=64) with B* = images*L = 16384 sequences.
    Algorithmic bytes of the main pass = qkv read once + sv|sve written once = 4*C*4*M (M = B* x L);
    statistics pass = q,k read once = C*4*M."""
    from medt_amd import _lib as ML
    from medt_amd.axial import AxialConfig, _desc, _params
    import lib as droplib
    lib = ML.lib()
    iters = int(os.environ.get("MEDT_ROOF_ITERS", iters))     # tuning aid: more launches per timing (kernel A/B runs)
    small_shape = images * L < 4096                           # the in-model shape (a few hundred sequences): latency-bound launches
    width = os.environ.get("MEDT_ROOF_AXIS", "w") != "h"      # tuning aid: the height-axis variant of the same shape
    layer = droplib.models.axialnet.AxialAttention_dynamic(C, C, groups=8, kernel_size=L, stride=1, width=width).to(device)
    layer.train()
    N, H, W = images, L, L
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn((N, C, H, W), generator=g).to(device)
    cfg = AxialConfig(8, 1 if width else 0, True, 1, layer.bn_qkv, layer.bn_similarity, layer.bn_output)
    desc = _desc(x, cfg, True)
    gates = (layer.f_qr, layer.f_kr, layer.f_sve, layer.f_sv)
    params = _params(cfg, layer.qkv_transform.weight, layer.relative, gates, True)
    sdt = torch.bfloat16 if cfg.act_dtype == 1 else torch.float32      # medt_amd.set_activation_dtype (--dtype bf16)
    e = 2 if cfg.act_dtype == 1 else 4
    qkv_raw = torch.empty((N, 2 * C, H, W), device=device, dtype=sdt)
    stacked = torch.empty((N, 2 * C, H, W), device=device, dtype=sdt)
    lse = torch.empty((N, 8, H, W), device=device)
    stats = torch.empty((lib.medt_axial_stats_floats(ctypes.byref(desc)),), device=device)
    ws_bytes = lib.medt_axial_workspace_bytes(ctypes.byref(desc))
    ws = torch.empty((ws_bytes,), device=device, dtype=torch.uint8)
    y = torch.empty((N, C, H, W), device=device)
    saved = ML.AxialSaved(qkv_raw.data_ptr(), stacked.data_ptr(), lse.data_ptr(), stats.data_ptr())
    stream = torch.cuda.current_stream().cuda_stream          # kernels are launched on torch's current stream
    ML.check(lib.medt_axial_layer_fwd(ctypes.byref(desc), ctypes.byref(params), x.data_ptr(), y.data_ptr(),
                                      ctypes.byref(saved), ws.data_ptr(), ws_bytes, stream), "layer_fwd")
    torch.cuda.synchronize()

    cold = {}

    def timed(fn, key=None):
        """Average launch duration by HIP events on the launch stream.  Two figures (round 6, profiles/r06_launch_spread.txt: the
        133 - 198 us spread of this kernel's launches inside one process is the clock governor -- an isolated launch runs at the
        boost clock, ~2 ms into a back-to-back burst the clock drops by ~20 % and steps back up in ~1 ms plateaus for ~10 ms):
        `cold` = 3 launches, then the next `iters` of the burst (rounds 1 - 5 measured this: the window sits on the dip);
        returned = the steady state: the burst continued for >= 25 ms, then >= 15 ms of launches timed."""
        def window(n):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            e1.synchronize()
            return e0.elapsed_time(e1) / n * 1e-3
        for _ in range(3):
            fn()
        t_cold = window(iters)
        if key:
            cold[key] = t_cold
        if small_shape:                                       # (latency-bound few-workgroup launches: no power transient to sit out)
            return t_cold
        window(max(3, min(400, int(0.025 / t_cold) + 1)))
        return window(max(iters, min(200, int(0.015 / t_cold) + 1)))

    t_main = timed(lambda: ML.check(lib.medt_axial_core_fwd(ctypes.byref(desc), ctypes.byref(params), ctypes.byref(saved),
                                                            ws.data_ptr(), ws_bytes, stream), "core_fwd"), "main")
    t_stats = timed(lambda: ML.check(lib.medt_axial_core_stats(ctypes.byref(desc), ctypes.byref(params), ctypes.byref(saved),
                                                               ws.data_ptr(), ws_bytes, stream), "core_stats"))
    # backward core (10*C*e*M over its two passes): one full layer backward fills the workspace
    # coefficients, then the two L x L passes are timed on their own
    dy = torch.randn((N, C, H, W), generator=g).to(device)
    dx = torch.empty_like(x)
    gsz = [2 * C * C, 2 * C, 2 * C, 24, 24, 2 * C, 2 * C, layer.relative.numel(), 4]
    gflat = torch.empty((sum(gsz),), device=device)
    gparts = list(torch.split(gflat, gsz))
    grads = ML.AxialGrads(*[t.data_ptr() for t in gparts[:8]], None)
    ML.check(lib.medt_axial_layer_bwd(ctypes.byref(desc), ctypes.byref(params), x.data_ptr(), None, dy.data_ptr(),
                                      ctypes.byref(saved), dx.data_ptr(), ctypes.byref(grads), ws.data_ptr(), ws_bytes,
                                      stream), "layer_bwd")
    torch.cuda.synchronize()
    # with the gates' gradients (the reference makes them trainable from epoch 10 on, train.py:169-171) ...
    t_bwd_gates = timed(lambda: ML.check(lib.medt_axial_core_bwd(ctypes.byref(desc), ctypes.byref(params), ctypes.byref(saved),
                                                                 dy.data_ptr(), ws.data_ptr(), ws_bytes, stream), "core_bwd"))
    # ... and without (requires_grad=False, axialnet.py:124-127: the state the timed training step is in): the same
    # layer without gate parameters (AxialAttention: gates == 1 in forward and backward) runs the instruction stream of the
    # frozen-gate step -- no gate-gradient accumulators in the sweep
    plain = droplib.models.axialnet.AxialAttention(C, C, groups=8, kernel_size=L, stride=1, width=width).to(device)
    plain.train()
    cfg_ng = AxialConfig(8, 1 if width else 0, True, 1, plain.bn_qkv, plain.bn_similarity, plain.bn_output)
    params_ng = _params(cfg_ng, plain.qkv_transform.weight, plain.relative, None, True)
    ML.check(lib.medt_axial_layer_fwd(ctypes.byref(desc), ctypes.byref(params_ng), x.data_ptr(), y.data_ptr(),
                                      ctypes.byref(saved), ws.data_ptr(), ws_bytes, stream), "layer_fwd")
    ML.check(lib.medt_axial_layer_bwd(ctypes.byref(desc), ctypes.byref(params_ng), x.data_ptr(), None, dy.data_ptr(),
                                      ctypes.byref(saved), dx.data_ptr(), ctypes.byref(grads), ws.data_ptr(), ws_bytes,
                                      stream), "layer_bwd")
    torch.cuda.synchronize()
    t_bwd = timed(lambda: ML.check(lib.medt_axial_core_bwd(ctypes.byref(desc), ctypes.byref(params_ng), ctypes.byref(saved),
                                                           dy.data_ptr(), ws.data_ptr(), ws_bytes, stream), "core_bwd"))
    M = N * H * W
    bytes_main = 4 * C * e * M
    bytes_stats = C * e * M
    flops_main = 7.0 * M * L * C
    # the launch: C = 16 -> ONE kernel, the bound-referenced four-rows-per-lane kernel (it redoes a row whose bound was too
    # loose itself since round 5: no flag memset, no repair launch behind it); C = 32 -> memset of the repair flag + the
    # bound-referenced one-row-per-lane kernel + the exact kernel's early exit;
    # MEDT_ROWS4=0 / MEDT_BOUND_PATH=0 select the other variants
    kname = ("attn_fwd4r_kernel<AXIS=%d,L=%d,EXACT=false,VEC=" + ("true" if width else "false") + ">" if C // 8 == 2 else "attn_fwd3_kernel<GP=%d,AXIS=%%d,L=%%d,EXACT=false>" % (C // 8))
    # bound: the main pass is limited by VALU issue, not by HBM (7 L C flop per position against 4 C e bytes = 28 flop/B at
    # L = 64: 224 TFLOP/s at 8 TB/s, above the 157 TFLOP/s fp32 vector peak; PMC: VALU busy 70 %, HBM traffic = the algorithmic
    # bytes) -- `frac` stays the HBM fraction SURVEY.md 8(d) defines, `valu_frac` is the fraction of the roof that binds
    roof = {"bound": "valu", "kernel": kname % (1 if width else 0, L),
            "shape": {"C": C, "G": 8, "L": L, "sequences": N * H, "bytes_per_launch": bytes_main,
                      "storage": "bf16" if e == 2 else "f32"},
            "achieved": bytes_main / t_main / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "frac": bytes_main / t_main / 1e9 / HBM_PEAK_GBPS, "traffic": None,
            "launch_ms": t_main * 1e3, "launch_ms_cold_burst": cold["main"] * 1e3,
            "timing": "launch_ms = steady state of a back-to-back burst (>= 25 ms warm, >= 15 ms timed, HIP events); launch_ms_cold_burst = "
                      "launches 4 .. %d of a burst from idle, the window of rounds 1 - 5, which sits on the clock governor's dip "
                      "(profiles/r06_launch_spread.txt)" % (3 + iters),
            "valu_tflops": flops_main / t_main / 1e12, "valu_peak_tflops": VALU_PEAK_TFLOPS,
            "valu_frac": flops_main / t_main / 1e12 / VALU_PEAK_TFLOPS, "hbm_frac": bytes_main / t_main / 1e9 / HBM_PEAK_GBPS,
            "stats_kernel": {"kernel": "sim_stats_rows_kernel" if width else "sim_stats_kernel",
                             "achieved": bytes_stats / t_stats / 1e9, "launch_ms": t_stats * 1e3,
                             "bytes_per_launch": bytes_stats, "frac": bytes_stats / t_stats / 1e9 / HBM_PEAK_GBPS},
            # SURVEY.md 8(d) "train fwd" = statistics pass + main pass: 6*C*e*M over both launches
            "train_fwd_core": {"bytes_per_launch": bytes_main + 2 * bytes_stats,
                               "achieved": (bytes_main + 2 * bytes_stats) / (t_main + t_stats) / 1e9,
                               "frac": (bytes_main + 2 * bytes_stats) / (t_main + t_stats) / 1e9 / HBM_PEAK_GBPS},
            # backward core = everything between bn_output's and bn_qkv's backward: the single sweep (attn_bwd_sweep_kernel) +
            # its closed-form bn_similarity corrections (attn_bwd_fix_kernel, attn_bwd_relfix_kernel, table sums, finalize);
            # gp > 4 / other lengths: the two generic passes (attn_bwd_stats_kernel + attn_bwd_kernel)
            "bwd_core": {"kernels": ("attn_bwd_sweep_kernel + attn_bwd_fix_kernel + attn_bwd_relfix_kernel + bwd_tables_kernel + "
                                     "sim_bwd_finalize_kernel" if C // 8 <= 4 and L in (32, 64, 128)
                                     else "attn_bwd_stats_kernel + attn_bwd_kernel"),
                         "bytes_per_launch": 10 * C * e * M,
                         "achieved": 10 * C * e * M / t_bwd / 1e9, "frac": 10 * C * e * M / t_bwd / 1e9 / HBM_PEAK_GBPS,
                         "launch_ms": t_bwd * 1e3, "gates": "frozen (as in the timed step)",
                         "with_gate_gradients": {"launch_ms": t_bwd_gates * 1e3,
                                                 "frac": 10 * C * e * M / t_bwd_gates / 1e9 / HBM_PEAK_GBPS}}}
    tf = os.path.join(ROOT, "profiles", "roofline_traffic.json")      # PMC-derived HBM bytes per launch, if collected
    if os.path.exists(tf) and (C, L, images) in ((16, 64, 256), (32, 128, 128)) and e == 4:
        try:
            tj = json.load(open(tf))
            if (C, L) == (32, 128):                           # the second shape's figures sit under their own key
                tj = tj.get("C32_L128", {})
            roof["traffic"] = tj.get("attn_fwd_bytes_per_launch")
            # NOT measured by this run: rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md) of the same
            # command, collected by scripts/collect_profiles.sh and committed
            roof["traffic_source"] = "profiles/roofline_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --roofline-only`, not this run)"
            if "attn_bwd_bytes_per_launch" in tj:
                roof["bwd_core"]["traffic"] = tj["attn_bwd_bytes_per_launch"]
        except Exception:
            pass
    return roof


# --------------------------------------------------------------------------- #
# CPU baseline leg: the oracle (a port of the reference's algorithm) on the host cores
# --------------------------------------------------------------------------- #
def cpu_baseline_leg(steps=8):
    from oracle import medt_oracle as O
    import lib as droplib
    O.set_fast_bn(True)                                  # aten's fused BatchNorm, like the reference's nn.BatchNorm
    torch.manual_seed(3000)
    # Host threads: MEDT_CPU_THREADS, default 16 -- the fastest setting in the sweeps on the 256-core MI355X hosts
    # (profiles/r02_cpu_thread_sweep.json, s/step at 8/16/32/64 threads: 2.04 / 1.89 / 2.73 / 5.95; round 1's box:
    # 1.68 / 1.96 / 2.79 / 6.37): the reference's tensors are a few MB, more OpenMP threads only add synchronisation.
    cores = min(os.cpu_count() or 1, int(os.environ.get("MEDT_CPU_THREADS", "16")))
    torch.set_num_threads(cores)
    log(f"cpu baseline: {cores} threads of {os.cpu_count()} cores")
    sd = droplib.models.axialnet.MedT(img_size=IMG, imgchan=3).state_dict()
    st = {k: v.clone() for k, v in sd.items()}
    leaves = []
    for k, v in st.items():
        if v.is_floating_point() and not k.endswith(("running_mean", "running_var")) and not k.endswith(("f_qr", "f_kr", "f_sve", "f_sv")):
            v.requires_grad_(True)
            leaves.append(v)
    opt = torch.optim.Adam(leaves, lr=1e-3, weight_decay=1e-5)
    x = torch.rand(PER_GPU_BATCH, 3, IMG, IMG)
    y = torch.randint(0, 2, (PER_GPU_BATCH, IMG, IMG))

    def step():
        out = O.medt(x, st, True)                       # literal 16-iteration patch loop, like the reference
        loss = O.log_nll_loss(out, y)
        opt.zero_grad()
        loss.backward()
        opt.step()
        for k in st:                                    # running stats were re-bound by the oracle; keep them detached
            if not st[k].requires_grad and st[k].is_floating_point():
                st[k] = st[k].detach()

    t0 = time.perf_counter()
    step()
    log(f"cpu baseline: warm-up step {time.perf_counter() - t0:.1f}s")
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = (time.perf_counter() - t0) / steps
    log(f"cpu baseline: {dt:.2f} s/step")
    O.set_fast_bn(False)
    return {"value": PER_GPU_BATCH / dt, "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{steps} training steps (1 warm-up) of MedT imgsize={IMG} bs={PER_GPU_BATCH} fp32 through oracle/medt_oracle.py "
                      f"(torch CPU, {cores} threads), fwd+CE+backward+Adam", "s_per_step": dt}


def box_probe():
    """Two probes of the GPU box beside the step time (scripts/ubench/*.hip; built by __graft_entry__.build()): boxes of the
    same SKU and reported clocks were seen at 2.2-2.3 and 3.6 ms/step.
    dependent_load_ns: latency of a dependent global load (L2 / Infinity Cache / HBM footprints).  Round 4: a 3.60 ms box read
    87 / 222 / 374 ns, the same as the 2.3 ms boxes -- this probe does NOT separate them.
    clock: the effective shader clock under light load (dependent v_fma / ds_read chains at 1 wave, 128 and 2048 workgroups):
    the kernels that are 2-5x slower on the slow boxes are the LDS / VALU chains on small grids; fast boxes read 2.61-2.65 ns
    per dependent FMA at every grid size."""
    import subprocess
    out = {}
    for key, name in (("dependent_load_ns", "load_latency"), ("clock", "clock_probe")):
        exe = os.path.join(ROOT, "scripts", "ubench", name + ".bin")
        src = os.path.join(ROOT, "scripts", "ubench", name + ".hip")
        try:
            if not os.path.exists(exe):
                subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", src, "-o", exe],
                               check=True, capture_output=True, timeout=120)
            line = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=60).stdout.strip().splitlines()[-1]
            j = json.loads(line)
            if key == "clock":                               # medians only: {grid: [ns per dependent FMA, ns per dependent LDS read]}
                j = {g: [v["ns_per_dependent_fma"]["median"], v["ns_per_dependent_lds_read"]["median"]] for g, v in j.items()}
                out[key] = j
            else:
                out.update(j)
        except Exception as e:                               # a probe, not the product: never fails the bench
            out[key + "_error"] = repr(e)[:200]
    return out


def baseline_config(args):
    """Which BASELINE.json configuration a (model, imgsize, per-GPU batch, dtype) line corresponds to."""
    key = (args.model, args.imgsize, args.batch, args.dtype)
    return {("MedT", 128, 4, "f32"): "configs[2] (per GPU: also configs[3]'s shard)",
            ("gatedaxialunet", 128, 8, "bf16"): "configs[1]",
            ("gatedaxialunet", 128, 4, "f32"): "configs[0]'s shape on the GPU",
            ("MedT", 256, 2, "f32"): "configs[4]'s per-GPU shard"}.get(key, "(not a BASELINE.json configuration)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="MedT")
    ap.add_argument("--imgsize", type=int, default=IMG)
    ap.add_argument("--batch", type=int, default=PER_GPU_BATCH, help="per-GPU batch")
    ap.add_argument("--eager", action="store_true", help="launch kernels from Python every step (no hipGraph replay)")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="storage of the attention layers' saved activations (bf16 = BASELINE.json configs[1]); "
                         "arithmetic and statistics are f32 either way")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--roofline-only", action="store_true", help="only the attention-kernel microbenchmark (for rocprofv3)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} (WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    force_dist = os.environ.get("MEDT_FORCE_DIST") == "1" and "RANK" in os.environ    # exercise the RCCL path on 1 GPU
    distributed = world > 1 or force_dist
    if distributed:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=device)
        log(f"rank {rank}/{world}: process group up, backend={dist.get_backend()}")

    import medt_amd
    medt_amd.set_activation_dtype(args.dtype)
    if args.roofline_only:
        # both SURVEY.md 8(d) shapes (their kernels are different template instances: the PMC passes tell them apart by name)
        print(json.dumps({"roofline": roofline_leg(device), "also": roofline_leg(device, C=32, L=128, images=128, iters=10)}))
        return

    from medt_amd import dp
    from medt_amd.optim import FlatAdam
    torch.manual_seed(3000)                              # (the reference seeds 3000 too, but at train.py:118-121, AFTER construction)
    model = build_model(args.model, args.imgsize, device)
    model.train()
    dp.broadcast_parameters(model)
    opt = FlatAdam(list(model.parameters()), lr=1e-3, weight_decay=1e-5)             # train.py:111-112
    g = torch.Generator().manual_seed(3000 + rank)
    x = torch.rand(args.batch, 3, args.imgsize, args.imgsize, generator=g).to(device)
    y = torch.randint(0, 2, (args.batch, args.imgsize, args.imgsize), generator=g).to(device)
    from medt_amd.trainer import TrainStep
    # fwd + LogNLLLoss (metrics.py:17-20) + bwd + gradient packing + fused Adam, captured into one hipGraph
    train_step = TrainStep(model, opt, medt_amd.cross_entropy, use_graph=not args.eager)

    def step():
        return train_step(x, y)

    log("model built; warm-up")
    for _ in range(args.warmup):
        step()
    log("warm-up done; timing")
    # EXACTLY --steps steps per timed window, barrier + synchronize on both sides, MAX over ranks.  The window is ~50 ms at
    # this step time, so one scheduler hiccup moves it by several percent: MEDT_BENCH_WINDOWS (default 5) back-to-back
    # windows are timed and the MEDIAN window is the one reported (ms_per_step x steps == that window's wall time).
    windows = []
    for _ in range(max(1, int(os.environ.get("MEDT_BENCH_WINDOWS", "5")))):
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            loss = step()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        el = time.perf_counter() - t0
        if distributed:
            t = torch.tensor([el], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = t.item()
        windows.append(el)
    elapsed = sorted(windows)[len(windows) // 2]
    final_loss = loss.item()
    log(f"{args.steps} steps per window; windows (ms): {[round(w * 1e3, 2) for w in windows]}; median {elapsed * 1e3:.2f}")

    result = {
        "metric": f"training images/sec ({args.model}, 3x{args.imgsize}x{args.imgsize})", "value": world * args.batch * args.steps / elapsed,
        "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "windows_ms": [round(w * 1e3, 3) for w in windows], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"{args.model} imgsize={args.imgsize} bs={args.batch}/GPU train step (fwd+CE+bwd+Adam), "
                               f"BASELINE.json {baseline_config(args)}" + ("" if world == 1 else f" x{world} data-parallel, flat-bucket all-reduce"),
                   "global_batch": world * args.batch, "parallelism": f"dp{world}"},
        "precision": ("f32 storage, f32 arithmetic" if args.dtype == "f32" else
                      "bf16 storage of the position-encoded attention layers' qkv / sv|sve activations, f32 arithmetic, "
                      "statistics, layer inputs/outputs and gradients"),
        "final_loss": final_loss, "hip_graph": not args.eager,
        "collective": (f"{dist.get_backend()} all_reduce(SUM) of the flat gradient bucket, "
                       + ("captured in the hipGraph (with the Adam launch behind it)" if getattr(train_step, "collective_in_graph", False)
                          else "outside the graph") if distributed else None),
        "collective_in_graph": bool(getattr(train_step, "collective_in_graph", False)) if distributed else None,
        # floats per all-reduce of each parameter group: two-branch networks have [global branch + trunk, local branch], the first
        # all-reduced where the global branch's backward ends (medt_amd.optim.TWO_BUCKETS)
        "gradient_buckets": [[hi - lo for lo, hi in g.bounds] for g in opt.groups],
    }
    if rank == 0 and world == 1:
        # BASELINE's second figure, "fwd ms/image": the eval-mode forward of reference test.py:106-119, replayed as a
        # hipGraph (medt_amd.trainer.InferStep, what test.py runs) at the bench batch and at test.py's batch of 1;
        # the eager launch-by-launch forward (host-bound) is reported beside it
        from medt_amd.trainer import InferStep
        model.eval()
        infer = InferStep(model)

        def time_fwd(fn, xin, reps):
            for _ in range(3):
                fn(xin)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):
                fn(xin)
            torch.cuda.synchronize()
            return (time.perf_counter() - t1) / reps / xin.shape[0] * 1e3

        with torch.no_grad():
            result["fwd_ms_per_image"] = time_fwd(infer, x, 50)
            result["fwd_ms_per_image_graph"] = result["fwd_ms_per_image"]
            # test.py's loop (reference test.py:106-119: loader batch size 1): since round 6 it gathers `--gather` (default 4)
            # loader items per replay -- eval mode, running statistics: the images do not interact -- so a batch-size-1 loader
            # costs one replay per FOUR images plus the concatenation; one image per replay is reported beside it
            singles = [x[k:k + 1].contiguous() for k in range(x.shape[0])]
            def gathered(xs):
                return infer(torch.cat(xs))
            for _ in range(3):
                gathered(singles)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(50):
                gathered(singles)
            torch.cuda.synchronize()
            result["fwd_ms_per_image_bs1"] = (time.perf_counter() - t1) / 50 / len(singles) * 1e3
            result["fwd_ms_per_image_bs1_single_replay"] = time_fwd(infer, x[:1].contiguous(), 50)
            result["fwd_ms_per_image_eager"] = time_fwd(model, x, 10)
            result["fwd_path"] = ("fwd_ms_per_image (= _graph): hipGraph replay of the eval-mode forward (InferStep) at the bench batch -- "
                                  "NOT comparable with BENCH_r01-r04's fwd_ms_per_image, which was the eager forward (now "
                                  "fwd_ms_per_image_eager: one Python-issued launch per kernel); _bs1: test.py's loop, batch-size-1 "
                                  f"loader items gathered {len(singles)} per replay (test.py --gather); _bs1_single_replay: one image per replay")
        log(f"eval fwd {result['fwd_ms_per_image']:.3f} ms/image replayed (bs {args.batch}), {result['fwd_ms_per_image_bs1']:.3f} for a "
            f"batch-size-1 loader gathered {len(singles)} per replay, {result['fwd_ms_per_image_bs1_single_replay']:.3f} one image per replay, "
            f"{result['fwd_ms_per_image_eager']:.3f} eager")
        if not args.no_roofline:
            result["roofline"] = roofline_leg(device)
            # SURVEY.md 8(d)'s second scaled shape (256-px inputs: C=32, gp=4, L=128), reported beside the headline one
            other = roofline_leg(device, C=32, L=128, images=128, iters=10)
            # (round 5: with its backward core -- attn_bwd_sweep_kernel<4,128,32>; layer2.0 of the 256-px networks ran the two generic
            #  passes at 0.4 % of the HBM peak before)
            result["roofline"]["also"] = [{**{k: other[k] for k in ("kernel", "shape", "achieved", "frac", "launch_ms", "valu_tflops")},
                                           "traffic": other.get("traffic"), "bwd_core": other["bwd_core"]}]
            # the same layer at the size the timed step runs it (MedT layer1: 4 images, 256 sequences): different kernel
            # variants are dispatched there (attn_fwd3_kernel<2,AXIS,64,EXACT=true>; the scaled shape above runs the
            # 4-rows-per-lane bound-referenced attn_fwd4r_kernel) and the launch is latency-, not bandwidth-bound
            small = roofline_leg(device, C=16, L=64, images=args.batch, iters=50)
            result["roofline"]["in_model_shape"] = {
                "shape": small["shape"], "fwd_kernel": "attn_fwd3_kernel<GP=2,AXIS=1,L=64,EXACT=true>",
                "fwd_launch_us": small["launch_ms"] * 1e3, "stats_launch_us": small["stats_kernel"]["launch_ms"] * 1e3,
                "bwd_core_launch_us": small["bwd_core"]["launch_ms"] * 1e3, "fwd_frac": small["frac"]}
            log("roofline leg done")
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline_leg()
            result["vs_cpu_baseline"] = result["value"] / result["cpu_baseline"]["value"]
    if rank == 0:
        # LAST: the probes run as child processes with their own HIP context; the eager eval forward measured right behind them
        # read 2.0-2.6 ms/image instead of 0.7 (this process's queues had been switched out)
        result["box_probe"] = box_probe()
        print(json.dumps(result))
    if dist.is_initialized():
        # Orderly teardown.  The captured hipGraph holds RCCL kernel nodes that reference the communicator: release the
        # graphs (and their private memory pool) and drain the device BEFORE the process group is destroyed, instead of
        # leaving the order to interpreter exit (round 3 saw the torchrun worker SIGABRT at exit about once in ten launches).
        train_step._graphs.clear()
        del train_step
        import gc
        gc.collect()
        torch.cuda.synchronize()
        dist.barrier()
        torch.cuda.synchronize()
        dist.destroy_process_group()
        log("process group destroyed")


if __name__ == "__main__":
    main()
