import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: torch.from_numpy(z[k]) if z[k].ndim else z[k].item() for k in z.files}


def load_json(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        return json.load(f)


def sub(d, prefix):
    return {k[len(prefix):]: v for k, v in d.items() if k.startswith(prefix)}


def rel_err(a, b):
    """max|a-b| / max|b| — the parity metric of SURVEY.md §8(d)."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    den = b.abs().max().item()
    return (a - b).abs().max().item() / (den if den > 0 else 1.0)


def assert_close(a, b, tol, what=""):
    assert tuple(a.shape) == tuple(b.shape), f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    e = rel_err(a, b)
    assert e <= tol, f"{what}: rel err {e:.3e} > {tol:.1e}"
