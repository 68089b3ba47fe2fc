"""Training harness of the pretrain step -- mirrors of the reference's `calculate_losses` and
`train_loop` (reference madeleine/utils/trainer.py:20-77 and :80-144): same signatures, same per-stain
mask / gate logic, same summation order [global, local, (intra)] per stain, same sentinel / skip
behaviour.  Host logic stays Python; the numeric work is in the loss callables handed in.

Two host-side differences that do not change results:
  * row selection uses index_select with indices computed on the CPU labels instead of boolean-mask
    indexing of device tensors (which forces a device sync per stain);
  * when the global loss is madeleine_amd.InfoNCE, all stains of the step go through ONE batched launch
    set (InfoNCE.batched) instead of one call per stain.
"""
import time

import numpy as np
import torch

from .loss import InfoNCE as _HipInfoNCE
from .utils import set_model_precision, smooth_rank_measure

DEVICE = torch.device("cuda" if torch.cuda.is_available() else "cpu")
HE_POSITION = 0
WHOLE_VIEW_POSITION = 0


def _rows(mask_cpu: torch.Tensor, device) -> torch.Tensor:
    return mask_cpu.nonzero(as_tuple=True)[0].to(device, non_blocking=True)


def _batched_global(loss_fn, stains, active, wsi_embs, symmetric):
    """All active stains through one InfoNCE.batched call -> dict stain_idx -> scalar loss."""
    he_all = wsi_embs["HE"]
    dev = he_all.device
    kmax = max(len(idx) for _, idx in active)
    d = he_all.shape[2]
    S = len(active)
    Q = he_all.new_zeros(S, kmax, d)
    P = he_all.new_zeros(S, kmax, d)
    cnts = []
    for s, (stain_idx, idx_cpu) in enumerate(active):
        idx = idx_cpu.to(dev, non_blocking=True)
        k = len(idx_cpu)
        Q[s, :k] = he_all[:, WHOLE_VIEW_POSITION, :, stain_idx].index_select(0, idx)
        P[s, :k] = wsi_embs[stains[stain_idx]][:, WHOLE_VIEW_POSITION, :].index_select(0, idx)
        cnts.append(k)
    cnt = torch.tensor(cnts, dtype=torch.int32).to(dev, non_blocking=True)
    losses = loss_fn.batched(Q, P, cnt, symmetric=symmetric)
    return {stain_idx: losses[s] for s, (stain_idx, _) in enumerate(active)}


def calculate_losses(STAINS, loss_fn_interMod, loss_fn_interMod_local, loss_fn_intraMod, wsi_embs, token_embs,
                     modality_labels_withoutHE, args):
    """trainer.py:20-77."""
    losses = []
    atleast_two_loss_flag = False
    labels = modality_labels_withoutHE.detach().cpu()

    active = []
    for stain_idx, stain in enumerate(STAINS):
        stain_mask = labels[:, stain_idx].bool()
        if stain_mask.sum().item() > 1:
            active.append((stain_idx, stain_mask.nonzero(as_tuple=True)[0]))

    batched = None
    if loss_fn_interMod and active:
        if args.global_loss != "info-nce":
            raise AssertionError("invalid global loss")
        if isinstance(loss_fn_interMod, _HipInfoNCE) and loss_fn_interMod.reduction == 'mean':
            batched = _batched_global(loss_fn_interMod, STAINS, active, wsi_embs, args.symmetric_cl)

    for stain_idx, idx_cpu in active:
        stain = STAINS[stain_idx]
        dev = wsi_embs["HE"].device
        idx = idx_cpu.to(dev, non_blocking=True)
        # Global loss
        if loss_fn_interMod:
            if batched is not None:
                global_loss = batched[stain_idx]
            else:
                HE_for_stain = wsi_embs["HE"][:, WHOLE_VIEW_POSITION, :, stain_idx].index_select(0, idx)
                stain_ind = wsi_embs[stain][:, WHOLE_VIEW_POSITION, :].index_select(0, idx)
                global_loss = loss_fn_interMod(query=HE_for_stain, positive_key=stain_ind, symmetric=args.symmetric_cl)
            losses.append(global_loss)
        # Local loss
        if loss_fn_interMod_local:
            HE_tokens = token_embs["HE"][:, :, :, stain_idx].index_select(0, idx)
            IHC_tokens = token_embs[stain].squeeze().index_select(0, idx)
            got_loss = loss_fn_interMod_local(HE_tokens, IHC_tokens, subsample=256)
            losses.append(got_loss * args.local_loss_weight)
        # Intra modality loss (views 1 and 2)
        if loss_fn_intraMod:
            he1 = wsi_embs["HE"][:, 1, :, stain_idx].index_select(0, idx)
            st1 = wsi_embs[stain][:, 1, :].index_select(0, idx)
            he2 = wsi_embs["HE"][:, 2, :, stain_idx].index_select(0, idx)
            st2 = wsi_embs[stain][:, 2, :].index_select(0, idx)
            losses.append(loss_fn_intraMod(query=he1, positive_key=he2, symmetric=args.symmetric_cl))
            losses.append(loss_fn_intraMod(query=st1, positive_key=st2, symmetric=args.symmetric_cl))
        atleast_two_loss_flag = True

    if len(losses) > 0:
        loss = sum(losses)
    else:
        loss = -1
        assert loss == -1 and not atleast_two_loss_flag, "Loss should be -1 if there are no losses to calculate"
    return loss, atleast_two_loss_flag


def train_loop(args, loss_fn_interMod, loss_fn_interMod_local, loss_fn_intraMod, ssl_model, epoch, dataloader, optimizer,
               scheduler_warmup, scheduler):
    """trainer.py:80-144."""
    n_views = 3 if loss_fn_intraMod else 1
    ssl_model.train()
    torch_precision = set_model_precision(args.precision)
    autocast_on = torch_precision in (torch.bfloat16, torch.float16)  # fp32/fp64: torch disables autocast anyway

    ep_loss = 0.
    fb_time = 0.
    all_embeds = []
    for b_idx, data in enumerate(dataloader):
        if epoch == 0 and b_idx == 0:
            print("Using precision:", torch_precision)
        s_fb = time.time()
        modality_labels = data['modality_labels']
        modality_labels_withoutHE = modality_labels[:, HE_POSITION + 1:]

        optimizer.zero_grad()
        with torch.amp.autocast(device_type="cuda", dtype=torch_precision, enabled=autocast_on):
            wsi_embs, token_embs = ssl_model(data, device=DEVICE, n_views=n_views)
            loss, atleast_two_loss_flag = calculate_losses(args.STAINS, loss_fn_interMod, loss_fn_interMod_local,
                                                           loss_fn_intraMod, wsi_embs, token_embs,
                                                           modality_labels_withoutHE, args)

        all_embeds.extend(wsi_embs['HE'][:, WHOLE_VIEW_POSITION, :, 0].detach().to(torch.float32).cpu().numpy())

        if not atleast_two_loss_flag:
            print("Skipping batch with only HE")
            continue

        loss.backward()
        optimizer.step()
        if epoch <= args.warmup_epochs:
            scheduler_warmup.step()
        else:
            scheduler.step()

        if (b_idx % 3) == 0:
            print(f"Loss for batch: {b_idx} = {loss:.3f}")
        ep_loss += loss.item()
        fb_time += time.time() - s_fb

    all_embeds_tensor = torch.Tensor(np.array(all_embeds))
    rank = smooth_rank_measure(all_embeds_tensor)
    return ep_loss, rank
