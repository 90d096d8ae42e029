"""ctypes binding of libmedt_hip.so (include/medt_abi.h).

There is NO fallback: if the library is missing or a call fails, this raises.
torch must be imported first so that the process has exactly one HIP runtime
(torch's bundled libamdhip64.so.7, same SONAME the library links against).
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL: provides the HIP runtime the library binds to)

from .build import LIB_PATH

_lib = None
ABI_VERSION = 9


class MedtError(RuntimeError):
    pass


class AxialDesc(C.Structure):
    _fields_ = [("N", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("G", C.c_int32),
                ("axis", C.c_int32), ("has_pos", C.c_int32), ("stride", C.c_int32), ("training", C.c_int32),
                ("bn_groups", C.c_int32), ("eps", C.c_float), ("momentum", C.c_float), ("out_relu", C.c_int32),
                ("gate_mode", C.c_int32), ("act_dtype", C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [("N", C.c_int32), ("Cin", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cout", C.c_int32),
                ("K", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32), ("has_bias", C.c_int32),
                ("has_bn", C.c_int32), ("has_res", C.c_int32), ("relu", C.c_int32), ("training", C.c_int32),
                ("bn_groups", C.c_int32), ("eps", C.c_float), ("momentum", C.c_float), ("lean", C.c_int32)]


class BnPtrs(C.Structure):
    _fields_ = [("weight", C.c_void_p), ("bias", C.c_void_p), ("running_mean", C.c_void_p),
                ("running_var", C.c_void_p), ("num_batches_tracked", C.c_void_p)]


class AxialParams(C.Structure):
    _fields_ = [("w_qkv", C.c_void_p), ("bn_qkv", BnPtrs), ("bn_similarity", BnPtrs), ("bn_output", BnPtrs),
                ("relative", C.c_void_p), ("f_qr", C.c_void_p), ("f_kr", C.c_void_p), ("f_sve", C.c_void_p),
                ("f_sv", C.c_void_p)]


class AxialSaved(C.Structure):
    _fields_ = [("qkv_raw", C.c_void_p), ("stacked", C.c_void_p), ("lse", C.c_void_p), ("stats", C.c_void_p)]


class AxialGrads(C.Structure):
    _fields_ = [("w_qkv", C.c_void_p), ("bn_qkv_weight", C.c_void_p), ("bn_qkv_bias", C.c_void_p),
                ("bn_sim_weight", C.c_void_p), ("bn_sim_bias", C.c_void_p), ("bn_out_weight", C.c_void_p),
                ("bn_out_bias", C.c_void_p), ("relative", C.c_void_p), ("gates", C.c_void_p)]


class BlockDesc(C.Structure):
    _fields_ = [("N", C.c_int32), ("C", C.c_int32), ("width", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("G", C.c_int32), ("training", C.c_int32), ("bn_groups", C.c_int32), ("eps", C.c_float),
                ("momentum", C.c_float)]


class BlockParams(C.Structure):
    _fields_ = [("w_down", C.c_void_p), ("bn1", BnPtrs), ("height", AxialParams), ("width", AxialParams),
                ("w_up", C.c_void_p), ("bn2", BnPtrs)]


class BlockSaved(C.Structure):
    _fields_ = [("z1", C.c_void_p), ("y1", C.c_void_p), ("stats1", C.c_void_p), ("height", AxialSaved),
                ("y_h", C.c_void_p), ("width", AxialSaved), ("y_w", C.c_void_p), ("z2", C.c_void_p),
                ("stats2", C.c_void_p)]


class BlockS2Params(C.Structure):
    _fields_ = [("blk", BlockParams), ("w_ds", C.c_void_p), ("bn_ds", BnPtrs)]


class BlockS2Saved(C.Structure):
    _fields_ = [("blk", BlockSaved), ("zd", C.c_void_p), ("yd", C.c_void_p), ("statsd", C.c_void_p)]


class BlockGrads(C.Structure):
    _fields_ = [("w_down", C.c_void_p), ("bn1_weight", C.c_void_p), ("bn1_bias", C.c_void_p), ("height", AxialGrads),
                ("width", AxialGrads), ("w_up", C.c_void_p), ("bn2_weight", C.c_void_p), ("bn2_bias", C.c_void_p)]


# symbol -> (restype, argtypes); this table is also what tests/test_abi.py checks against the header
SIGNATURES = {
    "medt_abi_version": (C.c_int, []),
    "medt_last_error": (C.c_char_p, []),
    "medt_queue_create": (C.c_void_p, []),
    "medt_queue_destroy": (C.c_int, [C.c_void_p]),
    "medt_queue_bind": (C.c_int, [C.c_void_p, C.c_void_p]),
    "medt_queue_pending": (C.c_size_t, [C.c_void_p]),
    "medt_queue_flush": (C.c_int, [C.c_void_p, C.c_void_p]),
    "medt_queue_flush2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "medt_queue_discard": (C.c_int, [C.c_void_p]),
    "medt_axial_stats_floats": (C.c_size_t, [C.POINTER(AxialDesc)]),
    "medt_axial_workspace_bytes": (C.c_size_t, [C.POINTER(AxialDesc)]),
    "medt_axial_layer_fwd": (C.c_int, [C.POINTER(AxialDesc), C.POINTER(AxialParams), C.c_void_p, C.c_void_p,
                                       C.POINTER(AxialSaved), C.c_void_p, C.c_size_t, C.c_void_p]),
    "medt_axial_layer_bwd": (C.c_int, [C.POINTER(AxialDesc), C.POINTER(AxialParams), C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.POINTER(AxialSaved), C.c_void_p, C.POINTER(AxialGrads), C.c_void_p,
                                       C.c_size_t, C.c_void_p]),
    "medt_axial_core_stats": (C.c_int, [C.POINTER(AxialDesc), C.POINTER(AxialParams), C.POINTER(AxialSaved),
                                        C.c_void_p, C.c_size_t, C.c_void_p]),
    "medt_axial_core_fwd": (C.c_int, [C.POINTER(AxialDesc), C.POINTER(AxialParams), C.POINTER(AxialSaved),
                                      C.c_void_p, C.c_size_t, C.c_void_p]),
    "medt_axial_core_bwd": (C.c_int, [C.POINTER(AxialDesc), C.POINTER(AxialParams), C.POINTER(AxialSaved), C.c_void_p,
                                      C.c_void_p, C.c_size_t, C.c_void_p]),
    "medt_wopos_block_workspace_bytes": (C.c_size_t, [C.POINTER(BlockDesc)]),
    "medt_wopos_block_fwd": (C.c_int, [C.POINTER(BlockDesc), C.POINTER(BlockParams), C.c_void_p, C.c_void_p,
                                       C.POINTER(BlockSaved), C.c_void_p, C.c_size_t, C.c_void_p]),
    "medt_wopos_block_s2_workspace_bytes": (C.c_size_t, [C.POINTER(BlockDesc)]),
    "medt_wopos_block_s2_fwd": (C.c_int, [C.POINTER(BlockDesc), C.POINTER(BlockS2Params), C.c_void_p, C.c_void_p,
                                          C.POINTER(BlockS2Saved), C.c_void_p, C.c_size_t, C.c_void_p]),
    "medt_wopos_block_bwd_workspace_bytes": (C.c_size_t, [C.POINTER(BlockDesc)]),
    "medt_wopos_block_bwd": (C.c_int, [C.POINTER(BlockDesc), C.POINTER(BlockParams), C.c_void_p, C.c_void_p, C.c_void_p,
                                       C.POINTER(BlockSaved), C.c_void_p, C.c_void_p, C.POINTER(BlockGrads), C.c_void_p,
                                       C.c_size_t, C.c_void_p]),
    "medt_conv_stats_floats": (C.c_size_t, [C.POINTER(ConvDesc)]),
    "medt_conv_workspace_bytes": (C.c_size_t, [C.POINTER(ConvDesc)]),
    "medt_conv_block_fwd": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(BnPtrs),
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "medt_conv_block_bwd": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p, C.c_void_p, C.POINTER(BnPtrs), C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "medt_gate_mlp_fwd": (C.c_int, [C.c_void_p] * 9 + [C.c_int] * 5 + [C.c_void_p]),
    "medt_gate_mlp_bwd": (C.c_int, [C.c_void_p] * 13 + [C.c_int] * 5 + [C.c_void_p]),
    "medt_up2x_relu_add_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "medt_up2x_relu_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "medt_patch_gather": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "medt_logo_merge_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p]),
    "medt_logo_merge_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p]),
    "medt_ce_partials": (C.c_size_t, [C.c_int, C.c_int]),
    "medt_ce_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                              C.c_void_p]),
    "medt_ce_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                              C.c_int, C.c_void_p]),
    "medt_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float,
                                 C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "medt_relu_mask": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "medt_seg_counts": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]),
}


def lib():
    """The loaded library (raises MedtError if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MedtError(f"{LIB_PATH} not found: build it with `python __graft_entry__.py` "
                            "(hipcc --offload-arch=gfx950); there is no CPU / eager fallback")
        l = C.CDLL(os.environ.get("MEDT_LIB_OVERRIDE", LIB_PATH))   # override: kernel experiments only
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        if l.medt_abi_version() != ABI_VERSION:
            raise MedtError("libmedt_hip.so ABI version mismatch")
        _lib = l
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        raise MedtError(f"{what} failed ({rc}): {lib().medt_last_error().decode()}")


def ptr(t):
    return None if t is None else t.data_ptr()
