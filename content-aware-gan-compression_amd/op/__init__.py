"""Drop-in name for the reference's `op` package (reference op/__init__.py:1-2)."""
from cagc.op import FusedLeakyReLU, fused_leaky_relu, upfirdn2d  # noqa: F401
