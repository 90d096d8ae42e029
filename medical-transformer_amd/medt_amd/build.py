"""Build libmedt_hip.so (hipcc, gfx950 only) in-tree.

The library is a plain C-ABI shared object (include/medt_abi.h); it is built with
hipcc directly -- no torch headers, no cmake.  `python -m medt_amd.build` or
`__graft_entry__.build()` calls this; the GPU box uses the prebuilt .so that
travels with the tree.
"""
from __future__ import annotations

import os
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))       # medical-transformer_amd/
REPO_ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libmedt_hip.so")
SOURCES = ["medt_api.hip", "pointwise.hip", "axial_core.hip", "conv.hip", "elementwise.hip", "axial_fast.hip", "conv_mfma.hip", "axial_small.hip", "conv_small.hip", "axial_stats.hip", "defer.hip", "axial_bwd.hip", "block_small.hip"]
HEADERS = ["medt_common.h", "medt_kernels.h", "axial_tiles.h", "sim_tables.h", "defer.h", "fin_inline.h", os.path.join(REPO_ROOT, "include", "medt_abi.h")]


def _stale() -> bool:
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


ASAN_LIB_PATH = os.path.join(PKG_DIR, "libmedt_asan.so")


def build_asan(verbose: bool = False) -> str:
    """libmedt_asan.so: the same sources with the HOST side under AddressSanitizer (-fsanitize=address -fno-gpu-sanitize: the
    descriptor validation, workspace carving, job recording and launch wrappers; device code is not instrumented).
    tests/test_abi_fuzz.py drives it with fuzzed descriptors in a subprocess that preloads the ASAN runtime."""
    if os.path.exists(ASAN_LIB_PATH):
        t = os.path.getmtime(ASAN_LIB_PATH)
        deps = [os.path.join(CSRC, s) for s in SOURCES] + [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
        if all(os.path.getmtime(d) <= t for d in deps):
            return ASAN_LIB_PATH
    return build(force=True, verbose=verbose, defines=("-fsanitize=address", "-fno-gpu-sanitize", "-shared-libsan", "-g"),
                 lib_path=ASAN_LIB_PATH, obj_dir="build_asan", link_flags=("-fsanitize=address", "-shared-libsan"))


ABLATE_LIB_PATH = os.path.join(PKG_DIR, "libmedt_ablate.so")


def build_ablate(verbose: bool = False) -> str:
    """libmedt_ablate.so: the same sources with -DMEDT_ABLATE -- the only build in which MEDT_SKIP=<kernel families> is honoured
    (timing experiments, scripts/r6_skip.sh loads it through MEDT_LIB_OVERRIDE; every result of such a run is garbage).  The
    product library has no such switch."""
    return build(force=True, verbose=verbose, defines=("-DMEDT_ABLATE",), lib_path=ABLATE_LIB_PATH, obj_dir="build_ablate")


AB_LIB_PATH = os.path.join(PKG_DIR, "libmedt_ab.so")


def build_ab(*defines: str, verbose: bool = False) -> str:
    """libmedt_ab.so: the same sources with compile-time A/B switches (-DMEDT_AB_...: the side of a kernel experiment that is NOT the
    product's) -- loaded through MEDT_LIB_OVERRIDE by the A/B scripts, so the product library itself carries no switch for it."""
    return build(force=True, verbose=verbose, defines=tuple(defines), lib_path=AB_LIB_PATH, obj_dir="build_ab")


def asan_runtime() -> str:
    """Path of the AddressSanitizer runtime a non-instrumented python has to LD_PRELOAD before loading libmedt_asan.so."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    out = subprocess.run([hipcc, "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True).stdout.strip()
    if out and os.path.isabs(out) and os.path.exists(out):
        return out
    import glob
    cands = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    return cands[0] if cands else ""


def build(force: bool = False, verbose: bool = True, defines=(), lib_path: str = None, obj_dir: str = "build",
          link_flags=()) -> str:
    """defines / lib_path / obj_dir: an instrumented second library next to the product one (scripts/phase_stamps.py
    builds libmedt_stamps.so with -DMEDT_STAMPS and loads it through MEDT_LIB_OVERRIDE)."""
    if lib_path is None:
        if not force and not _stale():
            return LIB_PATH
        lib_path = LIB_PATH
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(CSRC, obj_dir), exist_ok=True)
    # axial_bwd.hip: the SLP vectorizer pairs the sweep's FMAs into v_pk_fma_f32, which runs at the scalar rate on gfx950
    # and costs a third of the loop in v_mov register shuffles to line the operand pairs up
    # block_small.hip: packing pairs of output channels into v_pk_fma_f32 costs s_mov pairs for the scalar weight operands
    # (measured: 2.2007 vs 2.2096 ms/step)
    units = [(s, s.replace(".hip", ".o"), ["-fno-slp-vectorize"] if s in ("axial_bwd.hip", "block_small.hip") else [])
             for s in SOURCES]
    # the bandwidth-tuned attention kernels once more with bfloat16 storage as a compile-time constant
    units.append(("axial_fast.hip", "axial_fast_bf16.o", ["-DMEDT_FAST_BF16=1"]))
    for s, oname, extra in units:          # one hipcc per translation unit, in parallel
        o = os.path.join(CSRC, obj_dir, oname)
        objs.append(o)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *extra, *defines, "-I", os.path.join(REPO_ROOT, "include"),
               "-I", CSRC, "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append(subprocess.Popen(cmd))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *link_flags, *objs, "-o", lib_path]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return lib_path


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB_PATH)
