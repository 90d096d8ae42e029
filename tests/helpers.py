"""Shared test helpers: golden fixtures, seeded states, comparisons."""
from __future__ import annotations

import json
import os

import numpy as np
import torch

from oracle import medt_oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_manifest = None


def manifest():
    global _manifest
    if _manifest is None:
        with open(os.path.join(GOLDEN, "state_manifest.json")) as f:
            _manifest = json.load(f)
    return _manifest


def blank_state(model_name: str, S: int, chan: int = 3):
    """Zero-filled state_dict with the reference's keys/shapes/dtypes (from the committed manifest)."""
    ent = manifest()[f"{model_name}/{S}/{chan}"]["state"]
    return {k: torch.zeros(shape, dtype=getattr(torch, dt)) for k, shape, dt in ent}


def seeded_state(model_name: str, S: int, seed: int, chan: int = 3):
    st = O.randomize_state(blank_state(model_name, S, chan), seed)
    # flatten_index is data, not randomised: rebuild it the way the reference registers it (axialnet.py:132-135)
    for k in st:
        if k.endswith("flatten_index"):
            L = int(round(st[k].numel() ** 0.5))
            ar = torch.arange(L)
            st[k] = (ar.view(L, 1) - ar.view(1, L) + L - 1).reshape(-1)
    return st


def seeded_input(seed, N, C, S, classes=2):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(N, C, S, S, generator=g, dtype=torch.float32)
    y = torch.randint(0, classes, (N, S, S), generator=g)
    return x, y


def probe_vector(name: str, numel: int, seed: int) -> torch.Tensor:
    h = (sum(ord(c) * (i + 1) for i, c in enumerate(name)) + seed) % (2 ** 31)
    g = torch.Generator().manual_seed(h)
    return torch.randn(numel, generator=g, dtype=torch.float64)


NPROBE = 8


def probe_matrix(name: str, numel: int, seed: int) -> torch.Tensor:
    """(NPROBE, numel): the probes of tests/golden/make_golden.py (row 0 = probe_vector(name))."""
    return torch.stack([probe_vector(name if j == 0 else f"{name}#{j}", numel, seed) for j in range(NPROBE)])


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def rel_err(a, b):
    """max |a-b| / max |b|  (b = reference)."""
    a = torch.as_tensor(a, dtype=torch.float64).cpu()
    b = torch.as_tensor(b, dtype=torch.float64).cpu()
    denom = b.abs().max().item()
    return (a - b).abs().max().item() / (denom if denom > 0 else 1.0)


def param_names(model_name, S, chan=3):
    return [k for k, _ in manifest()[f"{model_name}/{S}/{chan}"]["params"]]
