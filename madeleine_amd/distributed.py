"""Data-parallel pretrain step: one process per GPU over RCCL (torch.distributed backend "nccl" on ROCm).

The reference scales with single-process nn.DataParallel (setup_components.py:185-187): the losses are
evaluated ONCE on the gathered global batch -- InfoNCE sees every other case as a negative, GOT's
thresholds are global min/max.  The MI355X-native equivalent keeps those semantics with one process per
GPU:

  * cases shard over ranks; encoder, pooling and token projection are rank-local;
  * parameter gradients: PyTorch DDP bucketed all-reduce (mean) on RCCL -- 20 MB fp32;
  * ONE autograd-aware all-gather per step of the packed slide embeddings [B_l, M, 512] (plus presence
    labels) forms the full negative set; every rank then evaluates the same global InfoNCE, so the backward
    of the gather needs NO collective: rank r keeps its own slice of the (replicated) gradient, scaled by W so
    that DDP's mean over ranks reproduces the single-process global-batch gradient exactly.
  * rank-local loss terms (GOT) are multiplied by W for the same reason.

All collectives work on gloo as well (CPU tests, world_size 2).
"""
import os
from typing import Dict, Optional, Sequence

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None):
    """torchrun-style init (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the environment). Returns (rank, world, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local_rank


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


class _AllGatherReplicatedLoss(torch.autograd.Function):
    """all_gather along dim 0 whose consumer is a loss evaluated identically on every rank.

    backward: grad_in = W * grad_out[own slice]  (== the reduce-scatter-sum of W identical gradients)."""

    @staticmethod
    def forward(ctx, x, group):
        W = dist.get_world_size(group)
        ctx.rank, ctx.W, ctx.n = dist.get_rank(group), W, x.shape[0]
        x = x.contiguous()
        out = x.new_empty((W * x.shape[0],) + tuple(x.shape[1:]))
        dist.all_gather_into_tensor(out, x, group=group)
        return out

    @staticmethod
    def backward(ctx, g):
        sl = g[ctx.rank * ctx.n:(ctx.rank + 1) * ctx.n]
        return sl * float(ctx.W), None


def all_gather_replicated(x: torch.Tensor, group=None) -> torch.Tensor:
    if world_size(group) == 1:
        return x
    return _AllGatherReplicatedLoss.apply(x, group)


def all_gather_labels(labels: torch.Tensor, device, group=None) -> torch.Tensor:
    """[B_l, M] presence labels of every rank -> [W*B_l, M] on the host (tiny; issued at step start)."""
    if world_size(group) == 1:
        return labels.cpu()
    x = labels.to(device=device, dtype=torch.float32).contiguous()
    out = x.new_empty((world_size(group) * x.shape[0], x.shape[1]))
    dist.all_gather_into_tensor(out, x, group=group)
    return out.cpu()


def gather_slide_embeddings(wsi_embs: Dict[str, torch.Tensor], modalities: Sequence[str], group=None):
    """Packs the per-modality slide embeddings [B_l,V,512] into one [B_l, M*V*512] payload, all-gathers it once
    and returns the global dict in the reference's shapes (HE expanded over M-1)."""
    if world_size(group) == 1:
        return wsi_embs
    M = len(modalities)
    parts = [wsi_embs[m][..., 0] if m == "HE" else wsi_embs[m] for m in modalities]      # each [B_l,V,512]
    Bl, V, D = parts[0].shape
    payload = torch.stack(parts, dim=1).reshape(Bl, M * V * D)
    full = all_gather_replicated(payload, group).view(-1, M, V, D)
    out = {}
    for i, m in enumerate(modalities):
        e = full[:, i]
        out[m] = e.unsqueeze(3).expand(-1, -1, -1, M - 1) if m == "HE" else e
    return out
