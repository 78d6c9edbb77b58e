"""Operator layer (L1) — same public names as the reference's `op` package (op/__init__.py:1-2)."""
from .fused_act import FusedLeakyReLU, fused_leaky_relu
from .upfirdn2d import upfirdn2d

__all__ = ["FusedLeakyReLU", "fused_leaky_relu", "upfirdn2d"]
