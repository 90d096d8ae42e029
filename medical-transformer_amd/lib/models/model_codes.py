"""MI355X-native drop-in for the gate variants of the reference's lib/models/model_codes.py (its experimental model
zoo; not imported by the reference's lib/models/__init__.py and unreachable from its CLI -- SURVEY.md 8(f) rank 4).

  AxialAttention_gated_sig   model_codes.py:215-313   the four gates enter through a sigmoid (f_sv initialised to 5.0):
                             same fused kernels as AxialAttention_dynamic with medt_axial_desc.gate_mode = 1
                             (sigmoid in a prologue launch, chain rule on the gate gradients).

AxialAttention_gated_data (:316-443, four gates per sequence from a two-layer MLP on the pooled input) is restated in
oracle/medt_oracle.py (gate_mode="data", pinned against the reference class) but has no kernel path yet: the
attention kernels take scalar gates.  Constructing it here raises.
"""
from .axialnet import _AxialAttentionBase

__all__ = ["AxialAttention_gated_sig", "AxialAttention_gated_data"]


class AxialAttention_gated_sig(_AxialAttentionBase):
    _gated = True
    _gate_init = (0.1, 0.1, 0.1, 5.0)         # f_qr, f_kr, f_sve, f_sv (model_codes.py:243-246)
    _gate_mode = 1


class AxialAttention_gated_data:
    def __init__(self, *a, **k):
        raise NotImplementedError("AxialAttention_gated_data (reference model_codes.py:316-443) needs per-sequence gates "
                                  "in the attention kernels; only its oracle restatement exists (oracle/medt_oracle.py)")
