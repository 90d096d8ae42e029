"""Where do the ~20 us of a fused small-layer forward kernel go?  Builds libmedt_stamps.so (-DMEDT_STAMPS: thread 0 of
workgroup (0,0) of wopos_small_fwd_kernel records the 100 MHz wall clock at its phase boundaries), runs one
AxialAttention_wopos forward per shape on cuda:0 and prints the time between the stamps.
    python scripts/phase_stamps.py --build     (here: hipcc cross-compiles)        python scripts/phase_stamps.py   (GPU box)"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "medical-transformer_amd")
STAMPS_LIB = os.path.join(PKG, "libmedt_stamps.so")
sys.path[:0] = [PKG, ROOT]

if "--build" in sys.argv:
    from medt_amd import build as b
    b.build(force=True, verbose=False, defines=("-DMEDT_STAMPS=1",), lib_path=STAMPS_LIB, obj_dir="build_stamps")
    print(STAMPS_LIB)
    sys.exit(0)

os.environ["MEDT_LIB_OVERRIDE"] = STAMPS_LIB
import torch  # noqa: E402
import lib as droplib  # noqa: E402
from medt_amd import _lib as L  # noqa: E402

NAMES = ["weights -> LDS, first input batch loaded", "projection (q|k|v rows -> LDS, qkv_raw stored)", "bn_qkv sums (wave per channel)",
         "bn_qkv finalise (double)", "normalise q|k|v", "logits + block sums", "bn_similarity finalise", "softmax + P.V (+ stores)",
         "bn_output sums", "bn_output finalise", "apply + pool + store y"]
dev = torch.device("cuda:0")
h = L.lib()
h.medt_debug_stamps.restype = ctypes.c_int
h.medt_debug_stamps.argtypes = [ctypes.c_void_p]
for (C, S, width) in [(64, 8, False), (32, 16, False), (32, 16, True), (128, 4, False), (128, 2, True)]:
    m = droplib.models.axialnet.AxialAttention_wopos(C, C, groups=8, kernel_size=S, stride=1, width=width).to(dev)
    m.train()
    m.bn_groups = 16
    x = torch.randn(64, C, S, S, device=dev)
    best = None
    for rep in range(5):
        with torch.no_grad():
            m(x)
        torch.cuda.synchronize()
        st = (ctypes.c_ulonglong * 16)()
        assert h.medt_debug_stamps(st) == 0
        d = [(st[i + 1] - st[i]) * 10 for i in range(11)]
        if best is None or sum(d) < sum(best):
            best = d
    print(f"C={C} map {S}x{S} axis={'w' if width else 'h'}: total {sum(best) / 1000:.2f} us (thread 0 of workgroup 0; kernel launch to first stamp not included)")
    for n, v in zip(NAMES, best):
        print(f"    {v / 1000:6.2f} us  {n}")

# ---- the one-launch AxialBlock_wopos forward (csrc/block_small.hip), same mechanism ----
from medt_amd import net  # noqa: E402
BNAMES = ["input tile + BatchNorm parameters -> LDS", "conv_down + bn1 + ReLU", "height: projection + bn_qkv", "height: logits, bn_similarity, softmax, P.V",
          "height: bn_output + tile", "width: projection + bn_qkv", "width: logits, bn_similarity, softmax, P.V", "width: bn_output + ReLU + tile",
          "conv_up + bn2 + identity + ReLU"]
h.medt_debug_block_stamps.restype = ctypes.c_int
h.medt_debug_block_stamps.argtypes = [ctypes.c_void_p]
blk = droplib.models.axialnet.AxialBlock_wopos(128, 64, groups=8, base_width=64, kernel_size=4).to(dev).train()
x = torch.randn(64, 128, 4, 4, device=dev).relu_()
best = None
for rep in range(5):
    with torch.no_grad():
        net.axial_block_forward(blk, x, 16)
    torch.cuda.synchronize()
    st = (ctypes.c_ulonglong * 32)()
    assert h.medt_debug_block_stamps(st) == 0
    d = [(st[i + 1] - st[i]) * 10 for i in range(9)]
    if best is None or sum(d) < sum(best):
        best = d
print(f"AxialBlock_wopos C=128 width=64 4x4 maps, one launch: total {sum(best) / 1000:.2f} us (lane 0 of workgroup 0)")
for n, v in zip(BNAMES, best):
    print(f"    {v / 1000:6.2f} us  {n}")

# ---- ... and its one-launch backward (MEDT_BLOCK_BWD=1), stamps 10 .. 24 of the same array ----
from medt_amd import block  # noqa: E402
if block.BWD_ENABLED:
    WN = ["[mask,] bn_output backward, tiles", "softmax + bn_similarity backward", "dq | dk | dv", "bn_qkv backward + tile",
          "projection dgrad (next phase's global reads in flight)"]
    GNAMES = ["loads, mask, bn2 backward, tile", "conv_up dgrad"] + ["width: " + n for n in WN] + ["height: " + n for n in WN] + \
             ["bn1 backward + tile, identity gradient loaded", "conv_down dgrad + identity + deposit"]
    best = None
    for rep in range(5):
        xg = x.clone().requires_grad_(True)
        y = net.axial_block_forward(blk, xg, 16)
        y.backward(torch.ones_like(y))
        torch.cuda.synchronize()
        st = (ctypes.c_ulonglong * 32)()
        assert h.medt_debug_block_stamps(st) == 0
        d = [(st[i + 1] - st[i]) * 10 for i in range(10, 24)]
        if best is None or sum(d) < sum(best):
            best = d
    print(f"AxialBlock_wopos backward, one launch: total {sum(best) / 1000:.2f} us (lane 0 of workgroup 0)")
    for n, v in zip(GNAMES, best):
        print(f"    {v / 1000:6.2f} us  {n}")
else:
    print("(one-launch block backward not enabled: MEDT_BLOCK_BWD=1)")
