"""Oracle restatement of the KD generator step (reference train.py:280-308 + :145-184 + :203-206),
'Output_Only' mode, LPIPS off, content mask supplied as a {0,1} tensor [B,1,H,W] (what
Get_Masked_Tensor, Util/content_aware_pruning.py:90-117, derives from the BiSeNet parsing).
TEST INFRASTRUCTURE (see oracle/__init__.py)."""
import torch
import torch.nn.functional as F

from .ref_model import discriminator_forward_ref, generator_forward_ref


def kd_generator_losses_ref(student_sd, teacher_sd, d_sd, zs, inject_index, mask, student_noise=None,
                            teacher_noise=None, kd_l1_lambda=3.0):
    """Returns (g_loss, kd_l1_loss, student_image).  Gradients flow into student_sd tensors only."""
    s_img = generator_forward_ref(student_sd, zs, inject_index=inject_index, noise=student_noise)       # :291
    fake_pred = discriminator_forward_ref(d_sd, s_img)                                                  # :293
    g_loss = F.softplus(-fake_pred).mean()                                                              # :203-206
    with torch.no_grad():
        t_img = generator_forward_ref(teacher_sd, zs, inject_index=inject_index, noise=teacher_noise)   # :151
    kd_l1 = kd_l1_lambda * torch.mean(torch.abs(t_img * mask - s_img * mask))                           # :156-164
    return g_loss, kd_l1, s_img


def adam_step_ref(params, grads, state, lr, betas, eps=1e-8):
    """torch.optim.Adam (no weight decay, no amsgrad) restated — reference train.py:528-532,308."""
    b1, b2 = betas
    state["t"] = state.get("t", 0) + 1
    t = state["t"]
    out = {}
    for k, p in params.items():
        g = grads[k]
        m = state.setdefault("m/" + k, torch.zeros_like(p))
        v = state.setdefault("v/" + k, torch.zeros_like(p))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / (1 - b2 ** t) ** 0.5).add_(eps)
        out[k] = p - (lr / (1 - b1 ** t)) * m / denom
    return out


def d_losses_ref(student_sd, d_sd, real_img, zs, inject_index, noise):
    """D step loss (reference train.py:241-262 + :187-191): softplus(-D(real)).mean() + softplus(D(G(z))).mean()."""
    fake = generator_forward_ref(student_sd, zs, inject_index=inject_index, noise=noise)
    fake_pred = discriminator_forward_ref(d_sd, fake)
    real_pred = discriminator_forward_ref(d_sd, real_img)
    return F.softplus(-real_pred).mean() + F.softplus(fake_pred).mean(), real_pred.mean(), fake_pred.mean()


def r1_ref(d_sd, real_img):
    """R1 penalty (reference train.py:194-200,264-278): returns (r1_loss, real_pred)."""
    x = real_img.detach().clone().requires_grad_(True)
    pred = discriminator_forward_ref(d_sd, x)
    (g,) = torch.autograd.grad(pred.sum(), x, create_graph=True)
    return g.pow(2).reshape(g.shape[0], -1).sum(1).mean(), pred


def path_reg_ref(student_sd, zs, inject_index, noise, pl_noise, mean_path_length=0.0):
    """Path-length regulariser (reference model.py:661-666, train.py:310-338): (path_loss, path_lengths, new mean)."""
    from .ref_model import path_lengths_ref
    img, latent = generator_forward_ref(student_sd, zs, inject_index=inject_index, noise=noise, return_latent=True)
    pl = path_lengths_ref(img, latent, pl_noise)
    mean = mean_path_length + 0.01 * (pl.mean() - mean_path_length)
    return (pl - mean).pow(2).mean(), pl, mean.detach(), img
