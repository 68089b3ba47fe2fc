"""Shared helpers for the tests (oracle side).  Test infrastructure."""
import os

import numpy as np
import torch

from oracle import recipe
from oracle import restatement as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODS5 = ["HE", "HER2", "PGR", "KI67", "ER"]


def golden(name):
    return np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))


def t(shape, key, lo=-1.0, hi=1.0):
    return torch.from_numpy(recipe.uniform(shape, key, lo, hi))


def recipe_params(n_mod, d_in, tag, stain_encoding=False, requires_grad=False):
    shapes = R.param_shapes(n_mod, d_in, 4, stain_encoding)
    sd = {k: torch.from_numpy(v) for k, v in recipe.state_dict_recipe(shapes, tag).items()}
    if requires_grad:
        for v in sd.values():
            v.requires_grad_()
    return sd


def rel_err(a, b):
    a = torch.as_tensor(np.asarray(a), dtype=torch.float64) if not torch.is_tensor(a) else a.detach().double().cpu()
    b = torch.as_tensor(np.asarray(b), dtype=torch.float64) if not torch.is_tensor(b) else b.detach().double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_rel(a, b, floor=None):
    """max |a-b| / max(|b|, floor); floor defaults to 1e-3 * max|b| (elementwise 'rel' with a sane floor)."""
    a = torch.as_tensor(np.asarray(a)).double() if not torch.is_tensor(a) else a.detach().double().cpu()
    b = torch.as_tensor(np.asarray(b)).double() if not torch.is_tensor(b) else b.detach().double().cpu()
    if floor is None:
        floor = 1e-3 * float(b.abs().max().clamp_min(1e-30))
    return float(((a - b).abs() / b.abs().clamp_min(floor)).max())
