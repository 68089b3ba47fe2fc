"""Host-side glue the training harness needs (mirrors of reference madeleine/utils/utils.py:124-201)."""
import random

import numpy as np
import torch


def set_model_precision(precision):
    """utils.py:124-144."""
    if precision == 'float64':
        return torch.float64
    if precision == 'float32':
        return torch.float32
    if precision == 'bfloat16':
        return torch.bfloat16
    raise ValueError(f"Invalid precision: {precision}")


def set_deterministic_mode(SEED, disable_cudnn=False):
    """utils.py:147-177 (seeds torch / python / numpy; MIOpen knobs are irrelevant to this path)."""
    torch.manual_seed(SEED)
    random.seed(SEED)
    np.random.seed(SEED)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(SEED)


def smooth_rank_measure(embedding_matrix, eps=1e-7):
    """exp(entropy of the normalised singular values), rounded to 2 decimals (utils.py:180-201). CPU, once per epoch."""
    _, S, _ = torch.svd(embedding_matrix)
    p = S / torch.norm(S, p=1) + eps
    p = p[:embedding_matrix.shape[1]]
    return round(torch.exp(-torch.sum(p * torch.log(p))).item(), 2)
