"""SURVEY.md section 5 (sanitizer / fuzz of the shim): fuzzed descriptors through the C ABI's host side -- in-process against
the product library, and against an AddressSanitizer build of the same sources in a subprocess.  No GPU needed: every call
carries NULL tensors, so it ends in the validation layer or in host-side size arithmetic."""
import json
import os
import subprocess
import sys

import helpers as H  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DRIVER = os.path.join(ROOT, "tests", "abi_fuzz_driver.py")


def test_descriptor_fuzz_in_process():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import abi_fuzz_driver as D
    from medt_amd import build
    build.build(verbose=False)
    c = D.run(1500, seed=7)
    assert c["axial_ok"] > 50 and c["conv_ok"] > 50, c          # the fuzz reaches accepted shapes, not only rejections
    assert c["entry_calls"] > 20000


def test_descriptor_fuzz_under_address_sanitizer():
    """The host side of every translation unit compiled with -fsanitize=address (medt_amd.build.build_asan), loaded into a
    python that preloads the ASAN runtime: a heap / stack / global overflow or a use-after-free in the validation, geometry,
    workspace carving or queue code aborts the subprocess with an AddressSanitizer report."""
    from medt_amd import build
    lib = build.build_asan()
    rt = build.asan_runtime()
    assert rt, "AddressSanitizer runtime of the ROCm clang not found"
    env = dict(os.environ, MEDT_LIB_OVERRIDE=lib, LD_PRELOAD=rt,
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=86:verify_asan_link_order=0")
    r = subprocess.run([sys.executable, DRIVER, "1500", "11"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    c = json.loads(r.stdout.strip().splitlines()[-1])
    assert c["axial_ok"] > 50 and c["conv_ok"] > 50 and c["queue_ops"] > 50, c
