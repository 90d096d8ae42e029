import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "medical-transformer_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("MPLBACKEND", "Agg")


def pytest_addoption(parser):
    parser.addoption("--emulate", action="store_true", default=False,
                     help="run the tests marked gpu on the CPU lane emulator instead of a GPU (tests/lane_emu, tests/emu_device.py: the "
                          "same product code, CPU tensors, libmedt_emu.so; slow -- select a few tests with -k)")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def emulating(request):
    """True under --emulate: tests that start their own processes pass the emulated device on."""
    return bool(request.config.getoption("--emulate"))


@pytest.fixture(scope="session")
def device(request):
    import torch
    if request.config.getoption("--emulate"):
        # e.g.  python -m pytest tests/test_model_gpu.py -m gpu --emulate -k "axialunet_S64"
        import ctypes
        import test_lane_emu as T
        from emu_device import emulated_device
        from medt_amd import _lib as L
        lib = ctypes.CDLL(T.build_emulator())
        for name, (res, args) in L.SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype, fn.argtypes = res, args
        with emulated_device(lib):
            yield torch.device("cpu")
        return
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    yield torch.device("cuda:0")
