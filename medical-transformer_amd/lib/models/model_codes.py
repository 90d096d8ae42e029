"""MI355X-native drop-in for the gate variants of the reference's lib/models/model_codes.py (its experimental model
zoo; not imported by the reference's lib/models/__init__.py and unreachable from its CLI -- SURVEY.md 8(f) rank 4).

  AxialAttention_gated_sig   model_codes.py:215-313   the four gates enter through a sigmoid (f_sv initialised to 5.0):
                             same fused kernels as AxialAttention_dynamic with medt_axial_desc.gate_mode = 1
                             (sigmoid in a prologue launch, chain rule on the gate gradients).
  AxialAttention_gated_data  model_codes.py:316-443   four gates PER SEQUENCE from a two-layer MLP on the
                             sequence-averaged input (fcn1, fcn2; :371-380): medt_gate_mlp_fwd/bwd produce the
                             (B*, 4) gate tensor and its backward, the attention kernels read it with gate_mode = 2.
  AxialBlock_gated_data      model_codes.py:619-659   AxialBlock_dynamic's dataflow around it.
"""
import torch.nn as nn

from medt_amd import ops as _ops

from .axialnet import _AxialAttentionBase, _AxialBlockBase

__all__ = ["AxialAttention_gated_sig", "AxialAttention_gated_data", "AxialBlock_gated_data"]


class AxialAttention_gated_sig(_AxialAttentionBase):
    _gated = True
    _gate_init = (0.1, 0.1, 0.1, 5.0)         # f_qr, f_kr, f_sve, f_sv (model_codes.py:243-246)
    _gate_mode = 1


class AxialAttention_gated_data(_AxialAttentionBase):
    _gate_mode = 2

    def _gate_modules(self, in_planes):       # registered between bn_output and relative, as in the reference (:345-347)
        self.fcn1 = nn.Linear(in_planes, in_planes)
        self.fcn2 = nn.Linear(in_planes, 4)
        self.pool = nn.AdaptiveAvgPool2d((1, 1))          # module-tree parity; the mean is part of medt_gate_mlp_fwd

    def _gates(self, x):
        return (_ops.gate_mlp(x, self.fcn1, self.fcn2, self.width), None, None, None)


class AxialBlock_gated_data(_AxialBlockBase):
    _attention = AxialAttention_gated_data
