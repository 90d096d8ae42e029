"""Per-kernel timing of the one-launch block kernels on cuda:0, all variants, in a few seconds (first GPU call of round 5):
forward (default / second-generation instantiation), forward + backward per stage vs one-launch backward.
    python scripts/block_kernels_bench.py              (spawns one process per MEDT_BLOCK_PK value: the library reads it once)
Times are per call of net.axial_block_forward (+ backward) on layer3_p.1's shape (64 images = 16 patch groups x 4, 128 channels,
4x4 maps), eager, back to back on one stream: kernel time + launch gaps, not the in-graph critical path (bench.py measures that)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "medical-transformer_amd"), ROOT]


def child():
    import torch
    import lib as droplib
    from medt_amd import block, net
    dev = torch.device("cuda:0")
    blk = droplib.models.axialnet.AxialBlock_wopos(128, 64, groups=8, base_width=64, kernel_size=4).to(dev).train()
    x = torch.randn(64, 128, 4, 4, device=dev).relu_()
    dout = torch.randn_like(x)

    def timed(fn, iters=300):
        for _ in range(30):
            fn()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            t0.record()
            for _ in range(iters):
                fn()
            t1.record()
            torch.cuda.synchronize()
            best = min(best, t0.elapsed_time(t1) * 1000.0 / iters)
        return best

    def fwd():
        with torch.no_grad():
            net.axial_block_forward(blk, x, 16)

    def fwd_bwd():
        xg = x.detach().requires_grad_(True)
        net.axial_block_forward(blk, xg, 16).backward(dout)

    pk = os.environ.get("MEDT_BLOCK_PK", "0")
    print(f"MEDT_BLOCK_PK={pk}  forward, one launch: {timed(fwd):7.2f} us per call")
    block.ENABLED = False
    print(f"MEDT_BLOCK_PK={pk}  forward, four per-stage launches: {timed(fwd):7.2f} us per call")
    block.ENABLED = True
    block.BWD_ENABLED = False
    print(f"MEDT_BLOCK_PK={pk}  forward (one launch) + backward per stage (6 launches + jobs): {timed(fwd_bwd, 100):7.2f} us per call")
    block.BWD_ENABLED = True
    print(f"MEDT_BLOCK_PK={pk}  forward (one launch) + backward one launch (+ jobs):           {timed(fwd_bwd, 100):7.2f} us per call")


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        for pk in ("0", "1"):
            env = dict(os.environ, MEDT_BLOCK_BWD="1", MEDT_BLOCK_PK=pk)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=env, check=False)
