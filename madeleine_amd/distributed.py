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
    """torchrun-style init (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from the environment). Returns (rank, world, local_rank).
    MADELEINE_DIST_BACKEND overrides the backend (e.g. gloo, to run several ranks on ONE GPU for debugging)."""
    backend = backend or os.environ.get("MADELEINE_DIST_BACKEND") or None
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


def _host_staged(x: torch.Tensor, group) -> bool:
    """gloo has no all_gather for device tensors: stage through the host (debug / single-GPU multi-process runs only;
    the production backend is nccl = RCCL, which takes device tensors directly)."""
    return x.is_cuda and dist.get_backend(group) == "gloo"


def _all_gather_cat(x: torch.Tensor, group=None) -> torch.Tensor:
    """[n, ...] on every rank -> [W*n, ...] (rank-major)."""
    W = dist.get_world_size(group)
    x = x.contiguous()
    if _host_staged(x, group):
        xc = x.cpu()
        out = xc.new_empty((W * xc.shape[0],) + tuple(xc.shape[1:]))
        dist.all_gather_into_tensor(out, xc, group=group)
        return out.to(x.device)
    out = x.new_empty((W * x.shape[0],) + tuple(x.shape[1:]))
    dist.all_gather_into_tensor(out, x, group=group)
    return out


def _all_reduce_sum(x: torch.Tensor, group=None) -> torch.Tensor:
    if _host_staged(x, group):
        xc = x.cpu()
        dist.all_reduce(xc, group=group)
        return xc.to(x.device)
    dist.all_reduce(x, group=group)
    return x


class _AllGatherReplicatedLoss(torch.autograd.Function):
    """all_gather along dim 0 whose consumer is a loss evaluated identically on every rank.

    backward: grad_in = W * grad_out[own slice]  (== the reduce-scatter-sum of W identical gradients)."""

    @staticmethod
    def forward(ctx, x, group):
        W = dist.get_world_size(group)
        ctx.rank, ctx.W, ctx.n = dist.get_rank(group), W, x.shape[0]
        return _all_gather_cat(x, group)

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
    x = labels.to(dtype=torch.float32).contiguous()
    if dist.get_backend(group) != "gloo":
        x = x.to(device)
    return _all_gather_cat(x, group).cpu()


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


# --------------------------------------------------------------------------------------------------
# local (GOT) loss with global-batch semantics
# --------------------------------------------------------------------------------------------------
_SIDE_STREAMS = {}


class _fan_out:
    """Run independent launches on side HIP streams and join them back into the current stream:
        with _fan_out(device, n) as lanes:
            with lanes(i): launch_i()
    Each lane first waits for the work already queued on the current stream; on exit the current stream waits for
    every lane.  Tensors allocated inside a lane are only consumed after the join (same allocator stream semantics
    as torch.cuda.stream + wait_stream).  On CPU tensors (gloo tests) it is a no-op."""

    def __init__(self, device, n):
        self.on = torch.device(device).type == "cuda" and n > 1
        self.device, self.n = device, n

    def __enter__(self):
        if self.on:
            key = (str(self.device), self.n)
            if key not in _SIDE_STREAMS:
                _SIDE_STREAMS[key] = [torch.cuda.Stream(device=self.device) for _ in range(self.n)]
            self.streams = _SIDE_STREAMS[key]
            self.main = torch.cuda.current_stream(self.device)
            self.used = set()
        return self._lane

    def _lane(self, i):
        import contextlib
        if not self.on:
            return contextlib.nullcontext()
        st = self.streams[i]
        st.wait_stream(self.main)
        self.used.add(i)
        return torch.cuda.stream(st)

    def __exit__(self, *exc):
        if self.on:
            for i in self.used:
                self.main.wait_stream(self.streams[i])
        return False


class _GOTMulti(torch.autograd.Function):
    """S GOT problems (one per stain) in ONE autograd node with global-batch thresholds.

    forward : local extrema of every problem -> one all-gather [S,6] -> min/max over ranks -> forwards
    backward: reverse sweeps of every problem -> one all-reduce of the extrema gradients [S,6] -> finish
    Ranks that own no case of a stain pass an empty (k = 0) problem: they still take part in both collectives.
    With world size 1 this is S independent reference-semantics GOT calls."""

    @staticmethod
    def forward(ctx, impl, group, *tensors):
        S = len(tensors) // 2
        probs = [(tensors[2 * s].contiguous(), tensors[2 * s + 1].contiguous()) for s in range(S)]
        dev = tensors[0].device
        inf = float("inf")
        ext = torch.stack([impl.extrema(V, Q) if V.shape[0] > 0 else
                           torch.tensor([inf, -inf] * 3, device=dev, dtype=tensors[0].dtype) for V, Q in probs])
        W = world_size(group)
        if W > 1:
            allx = _all_gather_cat(ext, group).view(W, ext.shape[0], 6)
            even = (torch.arange(6, device=dev) % 2 == 0)
            ext = torch.where(even, allx.amin(dim=0), allx.amax(dim=0))
        outs, states = [], []
        with _fan_out(dev, S) as lanes:   # stains are independent: one HIP stream each (k <= 32 workgroups per problem)
            for s, (V, Q) in enumerate(probs):
                if V.shape[0] == 0:
                    outs.append(torch.zeros(2, device=dev, dtype=tensors[0].dtype))
                    states.append(None)
                else:
                    with lanes(s):
                        o, st = impl.forward(V, Q, ext[s])
                    outs.append(o)
                    states.append(st)
        ctx.impl, ctx.group, ctx.states = impl, group, states
        ctx.shapes = [(V.shape, Q.shape) for V, Q in probs]
        return torch.stack(outs)

    @staticmethod
    def backward(ctx, d_outs):
        impl, states = ctx.impl, ctx.states
        dev = d_outs.device
        d_outs = d_outs.contiguous()
        parts = []
        with _fan_out(dev, len(states)) as lanes:
            for s, st in enumerate(states):
                if st is None:
                    parts.append(torch.zeros(6, device=dev, dtype=d_outs.dtype))
                else:
                    with lanes(s):
                        parts.append(impl.backward_begin(st, d_outs[s]))
        dmm = torch.stack(parts)
        if world_size(ctx.group) > 1:
            dmm = _all_reduce_sum(dmm, ctx.group)
        grads = []
        with _fan_out(dev, len(states)) as lanes:
            for s, st in enumerate(states):
                if st is None:
                    grads += [d_outs.new_zeros(ctx.shapes[s][0]), d_outs.new_zeros(ctx.shapes[s][1])]
                else:
                    with lanes(s):
                        grads += list(impl.backward_finish(st, dmm[s]))
        return (None, None) + tuple(grads)


def got_multi(problems, impl=None, group=None) -> torch.Tensor:
    """problems: list of (V, Q) token tensors [k_s, n_s, d] (already sub-sampled) -> [S, 2] = (WD sum, GWD sum)."""
    if impl is None:
        from .functional import HipGotImpl as impl  # noqa: N813
    flat = []
    for V, Q in problems:
        flat += [V, Q]
    return _GOTMulti.apply(impl, group, *flat)


_STEP = [0]


def calculate_losses_dp(STAINS, loss_fn_interMod, got_impl, wsi_embs, token_embs, modality_labels_withoutHE, args,
                        labels_global_withoutHE=None, group=None, subsample=256, shared_seed=0, use_local_loss=True):
    """Data-parallel counterpart of calculate_losses (trainer.py:20-77) with the reference's global-batch semantics.

    wsi_embs / token_embs / modality_labels_withoutHE are this rank's shard.  Returns (loss, flag) where `loss`
    is the tensor to call .backward() on under DDP (gradient mean over ranks): the replicated global InfoNCE
    plus W x (this rank's GOT sum); averaged over ranks its gradient equals the single-process global-batch
    gradient of  sum_stains [InfoNCE + w * GOT]  (SURVEY.md section 8(e))."""
    from .trainer import calculate_losses
    W = world_size(group)
    dev = wsi_embs["HE"].device
    labels_l = modality_labels_withoutHE.detach().cpu()
    labels_g = labels_global_withoutHE if labels_global_withoutHE is not None else all_gather_labels(labels_l, dev, group)
    mods = ["HE"] + list(STAINS)
    embs_g = gather_slide_embeddings(wsi_embs, mods, group)
    loss_g, flag = calculate_losses(STAINS, loss_fn_interMod, None, None, embs_g, None, labels_g, args)
    if not flag or not use_local_loss or got_impl is None:
        return loss_g, flag
    _STEP[0] += 1
    problems = []
    for s_idx, stain in enumerate(STAINS):
        k_g = int(labels_g[:, s_idx].bool().sum().item())
        if k_g <= 1:
            continue
        # reference: randperm(k_global)[:subsample] (loss.py:282) = the first min(k_g, subsample) tokens, any order
        if k_g <= subsample:
            tok_idx = torch.arange(k_g)
        else:
            gen = torch.Generator().manual_seed(int(shared_seed) * 1000003 + _STEP[0] * 131 + s_idx)
            tok_idx = torch.randperm(k_g, generator=gen)[:subsample]
        tok_idx = tok_idx.to(dev)
        rows = labels_l[:, s_idx].bool().nonzero(as_tuple=True)[0].to(dev)
        he = token_embs["HE"][:, :, :, s_idx].index_select(0, rows).index_select(1, tok_idx)
        st = token_embs[stain].index_select(0, rows).index_select(1, tok_idx)
        problems.append((he if he.dtype == torch.float64 else he.float(), st if st.dtype == torch.float64 else st.float()))
    if not problems:
        return loss_g, flag
    outs = got_multi(problems, got_impl, group)                    # [S,2]
    local = (outs[:, 1] + outs[:, 0]).sum() * args.local_loss_weight
    loss = (loss_g if torch.is_tensor(loss_g) else 0.0) + float(W) * local
    return loss, flag
