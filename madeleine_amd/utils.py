"""Host-side glue the training harness needs (mirrors of reference madeleine/utils/utils.py:27-201 and the local half of
madeleine/models/factory.py:17-41)."""
import json
import os
import pickle
import random
from argparse import Namespace

import numpy as np
import torch

DEVICE = torch.device("cuda" if torch.cuda.is_available() else "cpu")


def run_inference(ssl_model, val_dataloader, config=None, torch_precision=None, bags_per_launch=None):
    """Slide-embedding extraction loop (utils.py:27-66): eval mode, no gradients, full bags through `ssl_model.encode_he` under the
    configured precision; returns ({"embeds": [n,512] fp32 array, "slide_ids": [...]}, smooth rank of the embeddings).  The dataloader
    yields (feats [1,N,D], slide_ids) like the reference's SimpleDataset (wsi_dataset.py:102-124).  Forward-only use of the HIP path:
    nothing is saved for backward.
    Organisation (same embeddings, bit for bit, as one encode_he call per bag): up to `bags_per_launch` bags go through ONE launch set
    (MADELEINE.encode_he_bags: packed tokens + cu_seqlens; bags of at most 256 patches go alone, they take the small-M kernels), and
    the embeddings stay on the device until the loop ends -- the reference's per-bag `.cpu()` is a device synchronisation per slide,
    which leaves the GPU idle while the host prepares the next bag.
    `bags_per_launch` = None: 4 in fp32, where packing is bit-identical to one bag per call, and 1 under bf16 / fp16 autocast, where the
    Linear and gate kernels pick their tile from the packed row count: a slide's embedding then depends (at the bf16 rounding level,
    < 1e-2 relative) on which bags it is packed with, so extraction would not be reproducible across dataloader orders.  Pass
    bags_per_launch > 1 explicitly to trade that for throughput (2.2k instead of ~1k bags/s on 30,000-patch bags)."""
    ssl_model.eval()
    precision = torch_precision if torch_precision is not None else set_model_precision(config.precision)
    reduced = precision in (torch.bfloat16, torch.float16)      # for fp32 / fp64 torch disables autocast
    if bags_per_launch is None:
        bags_per_launch = 1 if reduced else 4
    batched = bags_per_launch > 1 and hasattr(ssl_model, "encode_he_bags")
    outs, slide_ids, pending = [], [], []

    def flush():
        if not pending:
            return
        with torch.amp.autocast(device_type="cuda", dtype=precision, enabled=reduced):
            if len(pending) == 1:
                emb = ssl_model.encode_he(pending[0], device=DEVICE)
            else:
                emb = ssl_model.encode_he_bags(pending, device=DEVICE)
        outs.append(emb.float())
        pending.clear()

    with torch.no_grad():
        for feats, ids in val_dataloader:
            slide_ids.append(ids[0])
            n_patches = feats.shape[-2]
            if not batched or n_patches <= 256:
                flush()
                pending.append(feats)
                flush()
                continue
            pending.append(feats)
            if len(pending) >= bags_per_launch:
                flush()
        flush()
        embeds = torch.cat(outs).cpu().numpy() if outs else np.zeros((0, 512), dtype=np.float32)
    return {"embeds": embeds, "slide_ids": slide_ids}, smooth_rank_measure(torch.Tensor(embeds))


def extract_slide_level_embeddings(args, val_dataloaders, ssl_model):
    """utils.py:68-90: run_inference over every validation dataloader; one `<RESULS_SAVE_PATH>/<dataset>.pkl` per dataset
    ({"embeds", "slide_ids"}), the smooth rank printed (and sent to wandb when args.log_ml and wandb is installed)."""
    for dataset_name, loader in val_dataloaders.items():
        print(f"\n* Extracting slide-level embeddings of {dataset_name}")
        results, rank = run_inference(ssl_model, loader, config=args)
        print("Rank for {} = {}".format(dataset_name, rank))
        if getattr(args, "log_ml", False):
            try:
                import wandb
                wandb.run.summary["{}_rank".format(dataset_name)] = rank
            except ImportError:
                pass
        os.makedirs(args.RESULS_SAVE_PATH, exist_ok=True)
        with open(os.path.join(args.RESULS_SAVE_PATH, f"{dataset_name}.pkl"), "wb") as f:
            pickle.dump(results, f)


def load_checkpoint(args, ssl_model, path_to_checkpoint=None):
    """utils.py:92-121: load `model.pt` (explicit path, or under args.RESULS_SAVE_PATH) into ssl_model; a state dict saved
    from nn.DataParallel / DDP ('module.' prefixes) is accepted."""
    path = path_to_checkpoint if path_to_checkpoint is not None else os.path.join(args.RESULS_SAVE_PATH, "model.pt")
    # plain state dicts load with the safe unpickler (torch >= 2.6's default, which is what the reference's torch.load gets)
    state = torch.load(path, map_location="cpu", weights_only=True)
    if any(k.startswith("module.") for k in state):
        state = {(k[7:] if k.startswith("module.") else k): v for k, v in state.items()}
        print('Model loaded by removing module in state dict...')
    ssl_model.load_state_dict(state)
    return ssl_model


def create_model_from_pretrained(local_dir: str, download: bool = False, device=None):
    """factory.py:17-41: (model, precision) from a released checkpoint directory holding `model_config.json` + `model.pt`.
    `download=True` first fetches MahmoodLab/madeleine with huggingface_hub into local_dir (as the reference always does);
    the default reads what is already there (no network on the training boxes)."""
    from .model import create_model
    if download:
        from huggingface_hub import snapshot_download
        os.makedirs(local_dir, exist_ok=True)
        print(f"* Downloading model at {local_dir}")
        snapshot_download(repo_id="MahmoodLab/madeleine", local_dir=local_dir)
    with open(os.path.join(local_dir, "model_config.json")) as f:
        cfg = Namespace(**json.load(f))
    model = create_model(cfg, device=device if device is not None else DEVICE,
                         checkpoint_path=os.path.join(local_dir, "model.pt"))
    return model, set_model_precision(cfg.precision)


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
