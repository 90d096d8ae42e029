"""Load the reference's lib/models/axialnet.py from /root/reference for oracle pinning.

TEST INFRASTRUCTURE.  Used only by tests/ and tests/golden/make_golden.py, and
only where the read-only reference checkout exists (this build container; it
does not exist on the GPU box).

`import lib` fails in this image because lib/__init__.py:1 pulls in
torchvision through lib/build_dataloader.py; the model file itself only needs
torch + matplotlib.  So the two model files are loaded by path under a private
package name (``_medt_reference``) that cannot collide with the product's own
drop-in ``lib`` package.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("MEDT_REFERENCE_ROOT", "/root/reference")
_PKG = "_medt_reference"


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "lib", "models", "axialnet.py"))


def load():
    """Returns the reference ``lib.models.axialnet`` module (cached)."""
    name = _PKG + ".models.axialnet"
    if name in sys.modules:
        return sys.modules[name]
    if not available():
        raise FileNotFoundError(REFERENCE_ROOT)
    os.environ.setdefault("MPLBACKEND", "Agg")
    old = sys.dont_write_bytecode
    sys.dont_write_bytecode = True           # the reference tree is read-only
    try:
        models_dir = os.path.join(REFERENCE_ROOT, "lib", "models")
        for pkg, path in ((_PKG, os.path.join(REFERENCE_ROOT, "lib")), (_PKG + ".models", models_dir)):
            m = types.ModuleType(pkg)
            m.__path__ = [path]
            m.__package__ = pkg
            sys.modules[pkg] = m
        for sub in ("utils", "axialnet"):
            full = f"{_PKG}.models.{sub}"
            spec = importlib.util.spec_from_file_location(full, os.path.join(models_dir, sub + ".py"))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[full] = mod
            spec.loader.exec_module(mod)
    finally:
        sys.dont_write_bytecode = old
    return sys.modules[name]


def load_model_codes():
    """The reference's lib/models/model_codes.py (experimental gate variants, unreachable from its CLI)."""
    load()
    full = f"{_PKG}.models.model_codes"
    if full in sys.modules:
        return sys.modules[full]
    old = sys.dont_write_bytecode
    sys.dont_write_bytecode = True
    try:
        spec = importlib.util.spec_from_file_location(full, os.path.join(REFERENCE_ROOT, "lib", "models", "model_codes.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[full] = mod
        spec.loader.exec_module(mod)
    finally:
        sys.dont_write_bytecode = old
    return mod


def load_metrics():
    """The reference's metrics.py (LogNLLLoss)."""
    name = _PKG + "_metrics"
    if name in sys.modules:
        return sys.modules[name]
    old = sys.dont_write_bytecode
    sys.dont_write_bytecode = True
    try:
        spec = importlib.util.spec_from_file_location(name, os.path.join(REFERENCE_ROOT, "metrics.py"))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
    finally:
        sys.dont_write_bytecode = old
    return mod


def factory(model_name: str):
    """CLI model name (train.py:95-102) -> reference factory function."""
    ax = load()
    return {"axialunet": ax.axialunet, "gatedaxialunet": ax.gated, "MedT": ax.MedT, "logo": ax.logo}[model_name]
