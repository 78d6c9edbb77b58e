"""Channel-mask surgery on a Generator state dict and the uniform remove list — the two pieces of the
reference's prune chain the benchmark needs to build its 70 %-pruned student from a full teacher
(Util/mask_util.py:11-121 `Mask_the_Generator`, Util/pruning_util.py:197-244).  Pure CPU dict work, runs once.

Layer numbering (Util/network_util.py:27-38): mask[0] = channels of the constant input (= Cin of conv1),
mask[l+1] = output channels of styled conv l (conv1 is l = 0, convs.n is l = n + 1).  ToRGB m (to_rgb1 is
m = 0) reads the output of styled conv 2m, i.e. mask[2m + 1]."""
import re

import numpy as np
import torch


def network_shape(sd):
    """[Cin of every styled conv ..., Cout of the last] — Util/network_util.py:27-38."""
    keys = ["conv1.conv.weight"] + [k for k in sd if k.startswith("convs.") and k.endswith(".conv.weight")]
    return [int(sd[k].shape[2]) for k in keys] + [int(sd[keys[-1]].shape[1])]


def uniform_remove_list(shape, ratio):
    """Util/pruning_util.py:233-244: int(C * ratio) channels removed per layer."""
    return [int(c * ratio) for c in shape]


def masks_from_scores(scores, shape, remove):
    """Keep-masks: drop the `remove[l]` lowest-scoring channels of layer l — Util/pruning_util.py:197-230."""
    masks = []
    for sc, c, r in zip(scores, shape, remove):
        m = np.ones(c, dtype=bool)
        if 0 < r < c:
            m[np.argsort(np.asarray(sc))[:r]] = False
        masks.append(m)
    return masks


def _styled_layer(name):
    if name.startswith("conv1."):
        return 0
    m = re.match(r"convs\.(\d+)\.", name)
    return int(m.group(1)) + 1 if m else None


def _rgb_layer(name):
    if name.startswith("to_rgb1."):
        return 0
    m = re.match(r"to_rgbs\.(\d+)\.", name)
    return int(m.group(1)) + 1 if m else None


def mask_generator_state_dict(sd, masks):
    """Slice every channel-bearing tensor of a Generator state dict by the keep-masks."""
    masks = [torch.as_tensor(np.asarray(m), dtype=torch.bool) for m in masks]
    out = {}
    for k, v in sd.items():
        v = v.detach().cpu()
        ls, lr = _styled_layer(k), _rgb_layer(k)
        if k == "input.input":
            v = v[:, masks[0]]
        elif ls is not None:
            if k.endswith("conv.weight"):
                v = v[:, masks[ls + 1]][:, :, masks[ls]]
            elif k.endswith("modulation.weight") or k.endswith("modulation.bias"):
                v = v[masks[ls]]
            elif k.endswith("activate.bias"):
                v = v[masks[ls + 1]]
        elif lr is not None:
            if k.endswith("conv.weight"):
                v = v[:, :, masks[2 * lr + 1]]
            elif k.endswith("modulation.weight") or k.endswith("modulation.bias"):
                v = v[masks[2 * lr + 1]]
        out[k] = v.clone()
    return out
