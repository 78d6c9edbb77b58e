"""cagc — MI355X-native StyleGAN2 generator / KD-retrain hot path.

Host side (Python on PyTorch-ROCm) of libcagc_hip.so.  Mirrors the reference's operator and model API
(`op.upfirdn2d`, `op.fused_leaky_relu`, `op.FusedLeakyReLU`, `model.Generator`, ...) so that the
reference's prune.py / train.py / Util callers drop in unchanged.

Dispatch rule — identical to the reference (op/fused_act.py:105, op/upfirdn2d.py:146): tensors on the CPU
take a composed-PyTorch path, tensors on a GPU take the hand-written HIP kernels.  There is NO fallback for
GPU tensors: if libcagc_hip.so is missing or a kernel reports an error, a RuntimeError is raised.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
