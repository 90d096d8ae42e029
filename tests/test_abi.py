"""The C-ABI library builds, loads, and exports every symbol include/medt_abi.h declares (no compute)."""
import ctypes
import os
import re

import pytest
import torch

import helpers as H  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "medt_abi.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(medt_[a-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def lib():
    from medt_amd import build, _lib
    build.build(verbose=False)          # hipcc cross-compiles for gfx950 without a GPU
    return _lib.lib()


def test_header_symbols_exported(lib):
    names = header_functions()
    assert "medt_axial_layer_fwd" in names and "medt_axial_layer_bwd" in names
    for n in names:
        assert hasattr(lib, n), f"{n} declared in medt_abi.h but not exported"


def test_binding_table_matches_header(lib):
    from medt_amd import _lib
    assert sorted(_lib.SIGNATURES) == header_functions()


def test_struct_layouts_match_header():
    """sizeof of the ctypes mirrors == what the C compiler lays out (checked via the documented field lists)."""
    from medt_amd import _lib
    assert ctypes.sizeof(_lib.AxialDesc) == 15 * 4
    assert ctypes.sizeof(_lib.ConvDesc) == 17 * 4
    assert ctypes.sizeof(_lib.BnPtrs) == 5 * 8
    assert ctypes.sizeof(_lib.AxialParams) == 8 + 3 * 40 + 5 * 8
    assert ctypes.sizeof(_lib.AxialSaved) == 4 * 8
    assert ctypes.sizeof(_lib.AxialGrads) == 9 * 8
    assert ctypes.sizeof(_lib.BlockDesc) == 10 * 4
    assert ctypes.sizeof(_lib.BlockParams) == 8 + 40 + 2 * (8 + 3 * 40 + 5 * 8) + 8 + 40
    assert ctypes.sizeof(_lib.BlockSaved) == 3 * 8 + 32 + 8 + 32 + 3 * 8
    assert ctypes.sizeof(_lib.BlockGrads) == 3 * 8 + 2 * 9 * 8 + 3 * 8


def test_descriptor_validation_no_gpu_needed(lib):
    from medt_amd import _lib
    d = _lib.AxialDesc(2, 16, 8, 8, 8, 0, 1, 1, 1, 1, 1e-5, 0.1, 0)
    assert lib.medt_axial_workspace_bytes(ctypes.byref(d)) > 0
    assert lib.medt_axial_stats_floats(ctypes.byref(d)) == 4 * (32 + 24 + 32)
    bad = _lib.AxialDesc(2, 24, 8, 8, 8, 0, 1, 1, 1, 1, 1e-5, 0.1, 0)      # group_planes = 3
    assert lib.medt_axial_workspace_bytes(ctypes.byref(bad)) == 0
    assert b"group_planes" in lib.medt_last_error()


def test_conv_descriptor_validation(lib):
    from medt_amd import _lib
    d = _lib.ConvDesc(2, 8, 16, 16, 128, 3, 1, 1, 0, 1, 0, 1, 1, 1, 1e-5, 0.1)
    assert lib.medt_conv_workspace_bytes(ctypes.byref(d)) > 0
    assert lib.medt_conv_stats_floats(ctypes.byref(d)) == 4 * 128
    bad = _lib.ConvDesc(2, 8, 16, 16, 128, 5, 1, 2, 0, 1, 0, 1, 1, 1, 1e-5, 0.1)
    assert lib.medt_conv_workspace_bytes(ctypes.byref(bad)) == 0


def test_block_descriptor_validation(lib):
    """medt_wopos_block_workspace_bytes: > 0 exactly for the shapes the fused block kernel is built for."""
    from medt_amd import _lib
    ok = _lib.BlockDesc(64, 128, 64, 4, 4, 8, 1, 16, 1e-5, 0.1)           # layer3_p.1-3 at BASELINE's batch size
    if os.environ.get("MEDT_BLOCK_FUSED", "1") != "0" and os.environ.get("MEDT_DISABLE_SMALL", "0") != "1":
        assert lib.medt_wopos_block_workspace_bytes(ctypes.byref(ok)) > 0
    for bad in (_lib.BlockDesc(32, 128, 64, 4, 4, 8, 1, 16, 1e-5, 0.1),   # 2 images per group
                _lib.BlockDesc(64, 128, 64, 8, 8, 8, 1, 16, 1e-5, 0.1),   # 8x8 maps
                _lib.BlockDesc(64, 64, 32, 4, 4, 8, 1, 16, 1e-5, 0.1)):   # other widths
        assert lib.medt_wopos_block_workspace_bytes(ctypes.byref(bad)) == 0
    assert lib.medt_wopos_block_fwd(ctypes.byref(ok), None, None, None, None, None, 0, None) < 0      # null arguments: refused
    fused_on = os.environ.get("MEDT_BLOCK_FUSED", "1") != "0" and os.environ.get("MEDT_DISABLE_SMALL", "0") != "1"
    # the one-launch backward and the 8x8-map kernels: default since round 5, MEDT_BLOCK_BWD=0 / MEDT_BLOCK8=0 switch them off
    ok8 = _lib.BlockDesc(64, 64, 32, 8, 8, 8, 1, 16, 1e-5, 0.1)            # layer2_p.1 at BASELINE's batch size
    want_bwd = fused_on and os.environ.get("MEDT_BLOCK_BWD", "1") != "0"
    want8 = fused_on and os.environ.get("MEDT_BLOCK8", "1") != "0"
    assert (lib.medt_wopos_block_bwd_workspace_bytes(ctypes.byref(ok)) > 0) == want_bwd
    assert (lib.medt_wopos_block_workspace_bytes(ctypes.byref(ok8)) > 0) == want8
    assert (lib.medt_wopos_block_bwd_workspace_bytes(ctypes.byref(ok8)) > 0) == (want8 and want_bwd)
    assert lib.medt_wopos_block_bwd(ctypes.byref(ok), None, None, None, None, None, None, None, None, None, 0, None) < 0


def test_single_hip_runtime(lib):
    maps = {line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64" in line}
    assert len(maps) == 1, maps


def test_product_rejects_cpu_tensors_loudly():
    import lib as droplib
    from medt_amd import MedtError
    m = droplib.models.axialnet.gated(img_size=32, imgchan=3)
    with pytest.raises(MedtError):
        m(torch.zeros(1, 3, 32, 32))
