"""Oracle restatement of the content-mask derivation (reference Util/content_aware_pruning.py:61-117), in numpy —
integer / index work plus dyadic bilinear weights, so the mask must be reproduced BIT-exactly.
TEST INFRASTRUCTURE (see oracle/__init__.py)."""
import numpy as np

PARSING_SIZE = 512                      # :72, :102
CHANNEL_MEAN = (0.485, 0.456, 0.406)    # :70
CHANNEL_STD = (0.229, 0.224, 0.225)     # :71


def _src(out_size, in_size, scale_factor):
    """align_corners=False source coordinates of F.interpolate(scale_factor=...) in fp32, as ATen computes them."""
    scale = np.float32(1.0 / scale_factor)
    src = scale * (np.arange(out_size, dtype=np.float32) + np.float32(0.5)) - np.float32(0.5)
    src = np.maximum(src, np.float32(0))
    i0 = np.minimum(src.astype(np.int64), in_size - 1)
    i1 = i0 + (i0 < in_size - 1)
    l1 = np.clip(src - i0.astype(np.float32), 0, 1).astype(np.float32)
    return i0, i1, np.float32(1) - l1, l1


def bilinear_resize_ref(x, scale_factor):
    """x [..., H, W] float32 -> [..., floor(H*sf), floor(W*sf)]   (F.interpolate bilinear, align_corners=False)."""
    H, W = x.shape[-2:]
    oh, ow = int(np.floor(H * scale_factor)), int(np.floor(W * scale_factor))
    y0, y1, hy0, hy1 = _src(oh, H, scale_factor)
    x0, x1, wx0, wx1 = _src(ow, W, scale_factor)
    top = wx0 * x[..., y0[:, None], x0[None, :]] + wx1 * x[..., y0[:, None], x1[None, :]]
    bot = wx0 * x[..., y1[:, None], x0[None, :]] + wx1 * x[..., y1[:, None], x1[None, :]]
    return (hy0[:, None] * top + hy1[:, None] * bot).astype(np.float32)


def parsing_input_ref(img):
    """Batch_Img_Parsing's preprocessing (:73-82): [N,3,S,S] in [-1,1] -> normalised [N,3,512,512]."""
    t = np.clip((img.astype(np.float32) + np.float32(1)) / np.float32(2), 0, 1)
    t = bilinear_resize_ref(t, PARSING_SIZE / img.shape[-1])
    mean = np.asarray(CHANNEL_MEAN, np.float32).reshape(1, 3, 1, 1)
    std = np.asarray(CHANNEL_STD, np.float32).reshape(1, 3, 1, 1)
    return ((t - mean) / std).astype(np.float32)


def parsing_ref(logits):
    """:87 — argmax over the class axis (first maximum)."""
    return logits.argmax(1)


def content_mask_ref(parsing, size):
    """Get_Masked_Tensor's mask (:102-107): (cls > 0) & (cls != 16) -> bilinear to `size` -> > 0.5; [N,1,size,size]."""
    keep = ((parsing > 0) & (parsing != 16)).astype(np.float32)
    m = bilinear_resize_ref(keep, size / parsing.shape[-1])
    return (m > np.float32(0.5)).astype(np.float32)[:, None]
