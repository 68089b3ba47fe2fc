"""Host-side glue the training harness needs (mirrors of reference madeleine/utils/utils.py:27-66 and :124-201)."""
import random

import numpy as np
import torch

DEVICE = torch.device("cuda" if torch.cuda.is_available() else "cpu")


def run_inference(ssl_model, val_dataloader, config=None, torch_precision=None):
    """Slide-embedding extraction loop (utils.py:27-66): eval mode, no gradients, one full bag per batch through
    `ssl_model.encode_he` under the configured precision; returns ({"embeds": [n,512] fp32 array, "slide_ids": [...]},
    smooth rank of the embeddings).  The dataloader yields (feats [1,N,D], slide_ids) like the reference's SimpleDataset
    (wsi_dataset.py:102-124).  Forward-only use of the HIP path: nothing is saved for backward."""
    ssl_model.eval()
    precision = torch_precision if torch_precision is not None else set_model_precision(config.precision)
    reduced = precision in (torch.bfloat16, torch.float16)      # for fp32 / fp64 torch disables autocast
    embeds, slide_ids = [], []
    with torch.no_grad():
        for feats, ids in val_dataloader:
            with torch.amp.autocast(device_type="cuda", dtype=precision, enabled=reduced):
                wsi_embed = ssl_model.encode_he(feats, device=DEVICE)
            embeds.extend(wsi_embed.float().cpu().numpy())
            slide_ids.append(ids[0])
    embeds = np.array(embeds)
    return {"embeds": embeds, "slide_ids": slide_ids}, smooth_rank_measure(torch.Tensor(embeds))


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
