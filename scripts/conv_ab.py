#!/usr/bin/env python
"""MFMA tile kernel vs direct VALU kernel for the 1x1 convolutions / qkv_transform shapes of the step.

Times medt_conv_block_fwd / _bwd (no BatchNorm, no bias: the bare convolution kernels) on the layer shapes of
gatedaxialunet / MedT at imgsize 128, once with MEDT_DISABLE_MFMA=1 (conv2d_fwd_kernel<1,*>, conv2d_bwd_data_kernel<1,*>)
and once with MEDT_FORCE_MFMA=1 (conv_mfma_fwd_kernel<1>, flipped-weight MFMA dgrad).  Each mode runs in its own process
(the switches are read once).  Prints one JSON object; scripts/collect_profiles.sh stores it as profiles/r02_conv_ab.json.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [  # N, Cin, Cout, H(=W): conv_down / conv_up / downsample / qkv_transform of the 128-px models (bs 4 / 8 / 64 patches)
    (4, 32, 64, 64), (4, 64, 32, 64), (4, 64, 128, 32), (4, 128, 64, 32), (8, 64, 128, 32), (4, 128, 128, 16),
    (4, 128, 256, 16), (4, 256, 128, 8), (4, 16, 32, 64), (4, 32, 64, 32), (64, 64, 128, 8), (64, 128, 64, 4),
]


def worker():
    sys.path[:0] = [ROOT, os.path.join(ROOT, "medical-transformer_amd")]
    import torch
    from medt_amd import ops
    dev = torch.device("cuda:0")
    REP = 20

    def graphed_us(fn):
        """Kernel time without the Python dispatch: REP calls captured into one hipGraph, replayed."""
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(3):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(REP):
                fn()
        for _ in range(3):
            g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            g.replay()
        e1.record()
        e1.synchronize()
        return e0.elapsed_time(e1) / 10 / REP * 1e3

    out = []
    for N, Cin, Cout, H in SHAPES:
        conv = torch.nn.Conv2d(Cin, Cout, 1, bias=False).to(dev)
        x = torch.randn(N, Cin, H, H, device=dev, requires_grad=True)
        dy = torch.randn(N, Cout, H, H, device=dev)

        def fwd():
            with torch.no_grad():
                ops.conv_block(x, conv)

        def fb():
            yy = ops.conv_block(x, conv)
            yy.backward(dy)
            x.grad = None
            conv.weight.grad = None

        out.append({"N": N, "Cin": Cin, "Cout": Cout, "HW": H * H, "positions": N * H * H,
                    "fwd_us": round(graphed_us(fwd), 2), "fwd_bwd_us": round(graphed_us(fb), 2)})
    print(json.dumps(out))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "worker":
        return worker()
    res = {}
    for mode, env in (("valu", {"MEDT_DISABLE_MFMA": "1"}), ("mfma", {"MEDT_FORCE_MFMA": "1"})):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "worker"], env=dict(os.environ, **env),
                           capture_output=True, text=True, timeout=600)
        if r.returncode != 0:
            raise SystemExit(r.stderr[-2000:])
        res[mode] = json.loads(r.stdout.strip().splitlines()[-1])
    rows = []
    for a, b in zip(res["valu"], res["mfma"]):
        rows.append({**{k: a[k] for k in ("N", "Cin", "Cout", "HW", "positions")}, "valu_fwd_us": a["fwd_us"],
                     "mfma_fwd_us": b["fwd_us"], "valu_fwd_bwd_us": a["fwd_bwd_us"], "mfma_fwd_bwd_us": b["fwd_bwd_us"],
                     "fwd_winner": "mfma" if b["fwd_us"] < a["fwd_us"] else "valu"})
    print(json.dumps({"what": "1x1 convolution (qkv_transform / conv_down / conv_up / downsample shapes): 20 calls captured into "
                              "one hipGraph, HIP-event timing of the replay / 20; fwd_bwd = fwd + dgrad + wgrad (+ slab reduction)",
                      "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
