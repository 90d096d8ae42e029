"""medt_amd -- MI355X-native runtime under the Medical-Transformer module surface.

`lib.models.axialnet` (the drop-in mirror of the reference's module surface)
binds its forward/backward to libmedt_hip.so through this package.
"""
from ._lib import MedtError, lib  # noqa: F401
from .axial import axial_attention, set_activation_dtype  # noqa: F401
from .ops import cross_entropy, seg_counts  # noqa: F401,E402
