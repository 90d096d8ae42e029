"""Random convolution-block configurations (channels, kernel 1 / 3 / 7, stride, padding, bias | BatchNorm (+ residual), ReLU, batch,
BatchNorm groups, map size, mode) through ops.conv_block on the CPU lane emulator against float64 torch -- tests/test_ops_gpu.py's
test_conv_block with random cases.   python scripts/emu_conv_hunt.py <seed> <count>      (no GPU)"""
import ctypes as C
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "medical-transformer_amd"), ROOT]
import torch  # noqa: E402
import test_lane_emu as T  # noqa: E402
import test_ops_gpu as TO  # noqa: E402
from emu_device import emulated_device  # noqa: E402
from medt_amd import _lib as L  # noqa: E402

lib = C.CDLL(T.build_emulator())
for name, (res, args) in L.SIGNATURES.items():
    fn = getattr(lib, name)
    fn.restype, fn.argtypes = res, args
rng = random.Random(int(sys.argv[1]))
n_ok = n_bad = 0
while n_ok + n_bad < int(sys.argv[2]):
    Cin, Cout = rng.choice([3, 8, 16, 32, 40, 64, 72, 128]), rng.choice([2, 8, 16, 32, 36, 64, 128])
    K = rng.choice([1, 1, 3, 3, 7])
    stride = rng.choice([1, 1, 2])
    pad = rng.choice([0, K // 2])
    has_bn = rng.random() < 0.7
    bias = (not has_bn) and rng.random() < 0.6
    has_res = has_bn and rng.random() < 0.4
    relu = rng.random() < 0.6
    N = rng.choice([1, 2, 3, 4, 8])
    groups = rng.choice([g for g in (1, 2, 4, N) if N % g == 0]) if has_bn else 1
    S = rng.choice([2, 4, 5, 8, 12, 16, 24])
    So = (S + 2 * pad - K) // stride + 1
    training = rng.random() < 0.7
    if So < 1 or (stride == 2 and K == 1 and S % 2) or (training and has_bn and (N // groups) * So * So < 2):
        continue
    case = (Cin, Cout, K, stride, pad, bias, has_bn, has_res, relu, N, S, groups)
    try:
        with emulated_device(lib):
            TO.test_conv_block(case, training, torch.device("cpu"))
        n_ok += 1
    except Exception as e:  # noqa: BLE001
        n_bad += 1
        print("FAIL", case, "train" if training else "eval", type(e).__name__, str(e)[:300].replace("\n", " "))
print("ok", n_ok, "bad", n_bad)
