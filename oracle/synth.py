"""Machine-independent synthetic inputs for fixtures that are too large to commit (integer hashing in numpy — no
dependence on any library's RNG stream).  TEST INFRASTRUCTURE (see oracle/__init__.py)."""
import numpy as np


def hash01(shape, seed):
    """Uniform-looking float32 in [0,1) from a 32-bit integer mix of the element index."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15 & 0xFFFFFFFF)
    i = (i ^ (i >> np.uint64(16))) * np.uint64(0x45D9F3B) & np.uint64(0xFFFFFFFF)
    i = (i ^ (i >> np.uint64(16))) * np.uint64(0x45D9F3B) & np.uint64(0xFFFFFFFF)
    i = i ^ (i >> np.uint64(16))
    return ((i >> np.uint64(8)).astype(np.float32) / np.float32(1 << 24)).reshape(shape)


def parsing_logits(B, P=512, NC=19, seed=1):
    """Stand-in for BiSeNet's output: a face-like class map (ellipse of kept classes, a band of class 16 = cloth at the
    bottom, background elsewhere) as logits 1.5 * one-hot, plus hash noise in [0, 2): the argmax is speckled, so the
    mask's resize + threshold sees every neighbourhood pattern."""
    yy, xx = np.meshgrid(np.arange(P), np.arange(P), indexing="ij")
    lg = hash01((B, NC, P, P), seed) * np.float32(2.0)
    for b in range(B):
        cls = np.zeros((P, P), np.int64)
        r = ((yy - P // 2 - 10 * b) / (0.39 * P)) ** 2 + ((xx - P // 2 + 7 * b) / (0.29 * P)) ** 2
        cls[r < 1.0] = 1 + (b % 15)
        cls[yy > int(0.86 * P)] = 16
        lg[b] += np.float32(1.5) * (np.arange(NC)[:, None, None] == cls[None]).astype(np.float32)
    return lg
