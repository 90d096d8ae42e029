"""Drop-in replacement for the reference's `lib` package (reference lib/__init__.py).

Only the model surface used by train.py:95-102 / test.py:90-97 is provided; the
reference's classification scaffolding (build_dataloader / build_model /
build_optimizer / Metric, dead code that hard-requires torchvision) is out of
scope.
"""
import os as _os
import sys as _sys

_pkg_root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
if _pkg_root not in _sys.path:            # make `medt_amd` importable when only `lib` was put on the path
    _sys.path.insert(0, _pkg_root)

from . import models  # noqa: E402,F401
