"""Checkpoint format of the reference training loop: ONE dict
`{'g', 'd', 'g_ema', 'g_optim', 'd_optim'}` of state dicts written by `torch.save` (train.py:443-452), named
`<iteration zero-filled to 6>.pt`; resume parses the iteration back out of the FILE NAME (train.py:541, App. D-9).
prune.py writes `{'g', 'd', 'g_ema'}` without optimiser state (prune.py:60-64) — accepted here as well.

Files written here load in the reference (`torch.load` + `Build_Generator_From_Dict`) and vice versa: the state-dict keys
and their order are the contract of cagc/model.py."""
import os

import torch

from . import model as M
from . import prune

KEYS = ("g", "d", "g_ema", "g_optim", "d_optim")


def _unwrap(m):
    return m.module if hasattr(m, "module") else m


def checkpoint_name(iteration):
    return f"{str(int(iteration)).zfill(6)}.pt"


def iteration_from_filename(path):
    """train.py:541 — `int(args.ckpt[-9:-3])`: the six digits before '.pt'."""
    return int(str(path)[-9:-3])


def optim_state_for_checkpoint(optim):
    """`optim.state_dict()` with every state tensor cloned: no two entries share storage.  `GraphedKDStep.optim` is a VIEW of one
    flat Adam (all `step` entries alias ONE counter, the moments are slices of one buffer); torch.save / load_state_dict preserve
    such aliasing, and a plain Adam resumed from it would advance the shared counter once per parameter per step (wrong bias
    correction).  Accepts an optimiser or a GraphedKDStep."""
    if hasattr(optim, "optim_state_dict"):
        return optim.optim_state_dict()
    sd = optim.state_dict()
    sd["state"] = {k: {n: (v.detach().clone() if torch.is_tensor(v) else v) for n, v in st.items()} for k, st in sd["state"].items()}
    return sd


def save_checkpoint(ckpt_dir, iteration, generator, discriminator, g_ema, g_optim=None, d_optim=None):
    """torch.save of the reference's dict (train.py:443-452) to `<ckpt_dir>/<iteration:06d>.pt`; returns the path.
    DDP / DataParallel wrappers are unwrapped (the reference saves `.module`'s state dict).  g_optim may be an optimiser or a
    GraphedKDStep; optimiser state is written de-aliased (`optim_state_for_checkpoint`)."""
    state = {"g": _unwrap(generator).state_dict(), "d": _unwrap(discriminator).state_dict(), "g_ema": _unwrap(g_ema).state_dict()}
    if g_optim is not None:
        state["g_optim"] = optim_state_for_checkpoint(g_optim)
    if d_optim is not None:
        state["d_optim"] = optim_state_for_checkpoint(d_optim)
    os.makedirs(ckpt_dir, exist_ok=True)
    path = os.path.join(ckpt_dir, checkpoint_name(iteration))
    torch.save(state, path)
    return path


def build_generator_from_dict(model_dict, size=256, latent=512, n_mlp=8):
    """Util/network_util.py:88-103 `Build_Generator_From_Dict`: channel shape read from the conv weights, then a NON-strict
    load (a missing or renamed key fails silently in the reference — here the mismatch is at least reported)."""
    gen = M.Generator(size, latent, n_mlp, generator_net_shape=prune.network_shape(model_dict))
    missing, unexpected = gen.load_state_dict(model_dict, strict=False)
    gen.load_report = {"missing": list(missing), "unexpected": list(unexpected)}
    return gen


def load_checkpoint(path, size=256, latent=512, n_mlp=8, channel_multiplier=2, device="cpu", g_optim=None, d_optim=None,
                    load_train_state=False):
    """train.py:484-489, 538-543: returns dict(generator, g_ema, discriminator, start_iter, raw).  Optimiser states are
    restored into the optimisers passed in when `load_train_state` (they must have been built over these modules'
    parameters, as in train.py:528-540 — pass them in a second call, or use `restore_optimizers`)."""
    ckpt = torch.load(path, map_location="cpu")
    gen = build_generator_from_dict(ckpt["g"], size, latent, n_mlp).to(device)
    g_ema = build_generator_from_dict(ckpt["g_ema"], size, latent, n_mlp).to(device)
    g_ema.eval()
    disc = M.Discriminator(size, channel_multiplier=channel_multiplier)
    disc.load_state_dict(ckpt["d"])
    disc = disc.to(device)
    start_iter = 0
    if load_train_state:
        restore_optimizers(ckpt, g_optim, d_optim)
        start_iter = iteration_from_filename(path) + 1
    for m in (gen, g_ema, disc):
        M.invalidate_caches(m)
    return {"generator": gen, "g_ema": g_ema, "discriminator": disc, "start_iter": start_iter, "raw": ckpt}


def restore_optimizers(ckpt, g_optim=None, d_optim=None):
    """g_optim may be a `GraphedKDStep`: its captured Adam graph holds the state tensors by address, so the saved state is
    copied into them (`load_optim_state`) instead of replacing them."""
    if g_optim is not None and hasattr(g_optim, "load_optim_state"):
        g_optim.load_optim_state(ckpt["g_optim"])
    elif g_optim is not None:
        g_optim.load_state_dict(ckpt["g_optim"])
    if d_optim is not None:
        d_optim.load_state_dict(ckpt["d_optim"])
