"""On-device content mask of the distillation loss — the reference's `Batch_Img_Parsing` and the mask half of
`Get_Masked_Tensor` (Util/content_aware_pruning.py:61-88, :102-107) without their host round trips
(`.type(torch.FloatTensor)` every step, a Python loop over the batch, `.squeeze()` that breaks batch 1).

    parsing_input(img)          [B,3,S,S] in [-1,1] -> the parsing net's input [B,3,512,512]      (:70-82)
    content_mask(logits, S)     parsing logits [B,19,512,512] -> {0,1} mask [B,1,S,S]             (:87, :102-107)
    teacher_content_mask(...)   the two around a user-supplied parsing net (BiSeNet is out of scope: its weights are not
                                obtainable offline — SURVEY §8-a row 14)

GPU tensors run csrc/content_mask.hip through the C ABI (`cagc_parsing_input`, `cagc_content_mask`); CPU tensors take the
composed-PyTorch formulation, as every op of this package does.  The mask is integer / dyadic arithmetic: both paths
are bit-identical to the reference (tests/golden/content_mask.npz)."""
import ctypes

import torch
from torch.nn import functional as F

from . import _lib

PARSING_SIZE = 512
CHANNEL_MEAN = (0.485, 0.456, 0.406)
CHANNEL_STD = (0.229, 0.224, 0.225)
BACKGROUND_CLASS = 0
EXCLUDED_CLASS = 16      # "cloth" in the CelebAMask-HQ label set the reference's BiSeNet was trained on

_MEAN3 = (ctypes.c_float * 3)(*CHANNEL_MEAN)
_STD3 = (ctypes.c_float * 3)(*CHANNEL_STD)


def _torch_scale(scale_factor):
    """The coordinate scale ATen derives from a user-given scale_factor: float(1 / scale_factor)."""
    return float(torch.tensor(1.0 / scale_factor, dtype=torch.float64).float())


def parsing_input(img, parsing_size=PARSING_SIZE):
    """clamp((img+1)/2, 0, 1) -> bilinear resize to parsing_size (align_corners=False) -> ImageNet normalisation."""
    B, C, S, S2 = img.shape
    assert C == 3 and S == S2, "parsing_input: [B,3,S,S] image expected"
    if img.is_cuda and img.dtype == torch.float32:
        x = img.detach().contiguous()
        out = torch.empty(B, 3, parsing_size, parsing_size, dtype=x.dtype, device=x.device)
        with _lib.on_device(x):
            _lib.call("cagc_parsing_input", _lib.ptr(out), _lib.ptr(x), B, S, parsing_size,
                      _torch_scale(parsing_size / S), ctypes.cast(_MEAN3, ctypes.c_void_p), ctypes.cast(_STD3, ctypes.c_void_p))
        return out
    t = ((img.detach() + 1) / 2).clamp(0, 1)
    t = F.interpolate(t, scale_factor=parsing_size / S, mode="bilinear", align_corners=False)
    mean = torch.tensor(CHANNEL_MEAN, dtype=t.dtype, device=t.device).view(1, 3, 1, 1)
    std = torch.tensor(CHANNEL_STD, dtype=t.dtype, device=t.device).view(1, 3, 1, 1)
    return (t - mean) / std


def content_mask(logits, size, excluded_class=EXCLUDED_CLASS):
    """logits [B,NC,P,P] -> mask [B,1,size,size] of {0,1}: argmax over classes, keep = (cls > 0) & (cls != excluded),
    bilinear resize to `size`, > 0.5.  No gradient flows (the reference's mask is a constant of the loss)."""
    B, NC, P, P2 = logits.shape
    assert P == P2
    if logits.is_cuda and logits.dtype == torch.float32:
        lg = logits.detach().contiguous()
        mask = torch.empty(B, 1, size, size, dtype=lg.dtype, device=lg.device)
        ws = torch.empty(_lib.query("cagc_content_mask_workspace", B, P), dtype=torch.float32, device=lg.device)
        with _lib.on_device(lg):
            _lib.call("cagc_content_mask", _lib.ptr(mask), _lib.ptr(ws), _lib.ptr(lg), B, NC, P, size, _torch_scale(size / P),
                      excluded_class)
        return mask
    cls = logits.detach().argmax(1)
    keep = ((cls > BACKGROUND_CLASS) & (cls != excluded_class)).unsqueeze(1).to(logits.dtype)
    m = F.interpolate(keep, scale_factor=size / P, mode="bilinear", align_corners=False)
    return (m > 0.5).to(logits.dtype)


def teacher_content_mask(teacher_img, parsing_net, parsing_size=PARSING_SIZE):
    """mask [B,1,S,S] from the teacher's image through `parsing_net` (callable returning logits, or a tuple whose first
    element is the logits — BiSeNet's convention, :85) — what KD_loss computes at train.py:155-158."""
    with torch.no_grad():
        out = parsing_net(parsing_input(teacher_img, parsing_size))
        logits = out[0] if isinstance(out, (tuple, list)) else out
        return content_mask(logits, teacher_img.shape[-1])
