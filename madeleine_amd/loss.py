"""Losses of the cross-stain pretrain step -- mirrors of the reference's `InfoNCE` and `GOT`
(reference madeleine/utils/loss.py:10-133 and :278-302), computed by libmadeleine_amd.so.
"""
import torch
from torch import nn

from . import functional as MF

__all__ = ['InfoNCE', 'info_nce', 'GOT', 'init_intra_wsi_loss_function']


def _validate(query, positive_key, negative_keys, negative_mode):
    """Argument checks of loss.py:67-89 (same messages)."""
    if query.dim() != 2:
        raise ValueError('<query> must have 2 dimensions.')
    if positive_key.dim() != 2:
        raise ValueError('<positive_key> must have 2 dimensions.')
    if negative_keys is not None:
        if negative_mode == 'unpaired' and negative_keys.dim() != 2:
            raise ValueError("<negative_keys> must have 2 dimensions if <negative_mode> == 'unpaired'.")
        if negative_mode == 'paired' and negative_keys.dim() != 3:
            raise ValueError("<negative_keys> must have 3 dimensions if <negative_mode> == 'paired'.")
    if len(query) != len(positive_key):
        raise ValueError('<query> and <positive_key> must must have the same number of samples.')
    if negative_keys is not None:
        if negative_mode == 'paired' and len(query) != len(negative_keys):
            raise ValueError("If negative_mode == 'paired', then <negative_keys> must have the same number of samples as <query>.")
    if query.shape[-1] != positive_key.shape[-1]:
        raise ValueError('Vectors of <query> and <positive_key> should have the same number of components.')
    if negative_keys is not None:
        if query.shape[-1] != negative_keys.shape[-1]:
            raise ValueError('Vectors of <query> and <negative_keys> should have the same number of components.')


def info_nce(query, positive_key, negative_keys=None, temperature=0.1, reduction='mean', negative_mode='unpaired',
             symmetric=False):
    _validate(query, positive_key, negative_keys, negative_mode)
    if negative_keys is not None:
        # In the reference this branch builds logits and then falls off the end of the function, returning None
        # (loss.py:93-110): it is not a usable code path, so it is not reproduced.
        raise NotImplementedError("explicit negative_keys: the reference branch (loss.py:93-110) never returns a loss")
    if reduction not in ('mean', 'sum', 'none'):
        raise ValueError("reduction must be 'mean', 'sum' or 'none' (F.cross_entropy's values, loss.py:124)")
    k, d = query.shape
    q, p = query.float().contiguous(), positive_key.float().contiguous()
    if d % 32:   # the similarity kernel works on 32-wide feature blocks: zero columns change neither norms nor cosines
        pad = 32 - d % 32
        q, p = torch.nn.functional.pad(q, (0, pad)), torch.nn.functional.pad(p, (0, pad))
    cnt = torch.full((1,), k, dtype=torch.int32, device=query.device)
    if reduction == 'none':
        return MF.info_nce_rows(q.unsqueeze(0), p.unsqueeze(0), cnt, temperature, symmetric)[0]
    loss = MF.info_nce_batched(q.unsqueeze(0), p.unsqueeze(0), cnt, temperature, symmetric)[0]
    return loss * k if reduction == 'sum' else loss


class InfoNCE(nn.Module):
    """Same constructor and call signature as the reference class (loss.py:10-64)."""

    def __init__(self, temperature=0.1, reduction='mean', negative_mode='unpaired'):
        super().__init__()
        self.temperature = temperature
        self.reduction = reduction
        self.negative_mode = negative_mode

    def forward(self, query, positive_key, negative_keys=None, symmetric=False):
        return info_nce(query, positive_key, negative_keys, temperature=self.temperature, reduction=self.reduction,
                        negative_mode=self.negative_mode, symmetric=symmetric)

    def batched(self, Q, P, cnt, symmetric=False):
        """S problems at once: Q,P [S,Kmax,D] padded, cnt int32 [S] -> loss [S] (mean reduction)."""
        return MF.info_nce_batched(Q.float().contiguous(), P.float().contiguous(), cnt, self.temperature, symmetric)


def init_intra_wsi_loss_function(config):
    """loss.py:138-157."""
    if config["intra_modality_mode_wsi"] in ("reconstruct_avg_emb", "reconstruct_masked_emb"):
        return nn.MSELoss()
    return InfoNCE(temperature=config["temperature"])


def GOT(v_, q_, subsample=None):
    """Graph optimal transport token alignment (loss.py:278-302): sum_b GW_b + sum_b WD_b.

    The sub-sample quirk of the reference is kept: indices are torch.randperm(v_.shape[0]) -- the (masked)
    BATCH size k, not the token count -- so the first n = min(k, subsample) tokens of each bag are used, in a
    random order that only changes summation order."""
    if subsample is not None:
        patch_indices = torch.randperm(v_.shape[0])[:subsample].to(v_.device)
        v_ = v_.index_select(1, patch_indices)
        q_ = q_.index_select(1, patch_indices)
    out = MF.got(v_.float().contiguous(), q_.float().contiguous())
    return out[1] + out[0]
