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


# ---------------------------------------------------------------------------------------------------
# content-aware saliency sweep (reference Util/content_aware_pruning.py:152-249 + prune.py:39-46), on device
# ---------------------------------------------------------------------------------------------------
def salt_pepper(mask, prob, generator=None):
    """Vectorised stand-in for the reference's per-pixel Python loop (:152-171): inside `mask` each pixel is hit with
    probability `prob` and takes the value -1 or +1 (shared by the 3 colour channels).  Returns (hit, sp) as float
    [B,1,H,W].  The reference draws from NumPy's global RNG pixel by pixel, so only the distribution can match."""
    hit = (mask > 0.5) & (torch.rand(mask.shape, device=mask.device, generator=generator) < prob)
    sp = torch.randint(0, 2, mask.shape, device=mask.device, generator=generator).float() * 2 - 1
    return hit.float(), sp * hit.float()


def saliency_modules(g):
    """[conv1] + convs + [to_rgbs[-1]] — the layers whose weight gradient is scored (:189-190)."""
    g = g.module if hasattr(g, "module") else g
    return [g.conv1] + list(g.convs) + [g.to_rgbs[-1]]


def batch_saliency_scores(generator, img, hit, sp):
    """One batch: loss = sum over hit pixels of |sp - img| (what `sum|noisy - img|` reduces to, :184), backward, and
    per layer mean |dL/dW| over (out-channel, ky, kx) -> one score per INPUT channel (:192-195).  Stays on device."""
    loss = (hit * (sp - img).abs()).sum()
    generator.zero_grad(set_to_none=True)
    loss.backward()
    scores = [m.conv.weight.grad.abs().mean(dim=[0, 1, 3, 4]) for m in saliency_modules(generator)]
    generator.zero_grad(set_to_none=True)
    return scores


def content_aware_scores(generator, n_sample, batch_size, noise_prob, mask_fn, device, latent_dim=512, rank=0, world=1,
                         rng=None, seed=None, noise_fn=None):
    """Sum over batches of the per-batch scores (prune.py:45-46).  With world > 1 the batches are dealt round-robin
    to the ranks and the score vectors are summed with ONE all-reduce at the end.
    `seed`: every batch draws its latents and salt-and-pepper pattern from its own generator (seed + batch index), so the
    result does not depend on how the batches are dealt to ranks; `noise_fn(batch_index, batch)` optionally supplies the
    per-layer noise maps (default: fresh noise, as prune.py)."""
    n_batch = max(1, n_sample // batch_size)
    sizes = [batch_size] * (n_batch - 1) + [batch_size + n_sample % batch_size]
    total = None
    for idx, b in enumerate(sizes):
        if idx % world != rank:
            continue
        if seed is not None:
            rng = torch.Generator(device=device).manual_seed(int(seed) + idx)
        z = torch.randn(b, latent_dim, device=device, generator=rng)
        img = generator([z], noise=noise_fn(idx, b)) if noise_fn is not None else generator([z])
        hit, sp = salt_pepper(mask_fn(img.detach()), noise_prob, rng)
        sc = batch_saliency_scores(generator, img, hit, sp)
        total = sc if total is None else [a + c for a, c in zip(total, sc)]
    if total is None:
        total = [torch.zeros(m.conv.weight.shape[2], device=device) for m in saliency_modules(generator)]
    if world > 1:
        import torch.distributed as dist
        flat = torch.cat(total)
        dist.all_reduce(flat)
        out, off = [], 0
        for t in total:
            out.append(flat[off:off + t.numel()])
            off += t.numel()
        total = out
    return total
