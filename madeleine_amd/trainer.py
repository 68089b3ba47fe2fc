"""Host side of the pretrain step: `calculate_losses` and `train_loop` with the reference's signatures and
behaviour (reference madeleine/utils/trainer.py:20-77 and :80-144) -- per-stain presence gating (a stain takes
part only when more than one case of the batch has it), the term order [global, local, intra x2] per stain, the
(-1, False) sentinel for an H&E-only batch, the skip / scheduler / print behaviour of the loop.  The numeric work
is in the loss callables handed in (madeleine_amd.InfoNCE / GOT, or any callables with the same signatures).

Organisation differs from the reference where it costs device syncs or launches:
  * rows of a stain are picked with index_select on indices derived from the CPU label matrix (boolean-mask
    indexing of device tensors forces a sync per stain);
  * when the global loss is madeleine_amd.InfoNCE, all participating stains go through ONE batched launch set.
"""
import os
import time
from typing import Dict, List, NamedTuple, Optional

import numpy as np
import torch

from .loss import GOT as _GOT

from .functional import h2d
from .loss import InfoNCE as _HipInfoNCE
from .utils import set_model_precision, smooth_rank_measure

DEVICE = torch.device("cuda" if torch.cuda.is_available() else "cpu")
HE_POSITION = 0
WHOLE_VIEW_POSITION = 0


# A/B switch (tools/runs/r05_t.sh): MADELEINE_LOSS_GATHER_ALWAYS=1 restores the unconditional gathers / zero-padded problem tensors
_SKIP_IDENTITY_GATHERS = not os.environ.get("MADELEINE_LOSS_GATHER_ALWAYS")


class _Participant(NamedTuple):
    """A stain that enters the loss of this batch, with the batch rows (cases) that carry it."""
    column: int            # position in STAINS (= column of modality_labels_withoutHE)
    name: str
    rows_cpu: torch.Tensor


def _participants(STAINS, labels_cpu: torch.Tensor) -> List[_Participant]:
    out = []
    for column, name in enumerate(STAINS):
        rows = labels_cpu[:, column].bool().nonzero(as_tuple=True)[0]
        if rows.numel() > 1:   # trainer.py:28 -- a single case cannot form a contrastive pair
            out.append(_Participant(column, name, rows))
    return out


def _all_cases(part: _Participant, batch: int) -> bool:
    """Every case of the batch carries this stain: the row gather is the identity (rows_cpu is ascending and unique)."""
    return _SKIP_IDENTITY_GATHERS and part.rows_cpu.numel() == batch


def _take_rows(x, part: _Participant, rows):
    """x.index_select(0, rows) -- skipped when every case participates (same values; the gather and, in the backward, its zero fill +
    index_add are ~6 launches of a few microseconds each per stain and side, on a device that is never idle)."""
    return x if _all_cases(part, x.shape[0]) else x.index_select(0, rows(part) if callable(rows) else rows)


def _pair(wsi_embs, part: _Participant, rows, view: int):
    """(H&E embedding matched to this stain, stain embedding) of view `view` for the participating cases.  `rows`: the device index
    tensor, or a callable part -> tensor (a _RowIndex: only called when a gather is needed)."""
    he = _take_rows(wsi_embs["HE"][:, view, :, part.column], part, rows)
    st = _take_rows(wsi_embs[part.name][:, view, :], part, rows)
    return he, st


class _RowIndex:
    """Device copies of the participants' row lists, uploaded at most once per calculate_losses call and only when a gather needs them."""

    def __init__(self, dev):
        self.dev, self._on_dev = dev, {}

    def __call__(self, part: _Participant):
        if part.column not in self._on_dev:
            self._on_dev[part.column] = h2d(part.rows_cpu, self.dev)
        return self._on_dev[part.column]


def _global_terms_batched(criterion, parts: List[_Participant], wsi_embs, symmetric, rows_of) -> Dict[int, torch.Tensor]:
    """All participating stains in one InfoNCE.batched call: padded [S, k_max, d] problems + live-row counts."""
    he_all = wsi_embs["HE"]
    dev, d = he_all.device, he_all.shape[2]
    k_max = max(p.rows_cpu.numel() for p in parts)
    pairs = [_pair(wsi_embs, part, rows_of, WHOLE_VIEW_POSITION) for part in parts]
    if _SKIP_IDENTITY_GATHERS and all(p.rows_cpu.numel() == k_max for p in parts):   # nothing to pad: no zero fill, no slice assignments (and none of their backwards)
        Q, P = torch.stack([q for q, _ in pairs]), torch.stack([p for _, p in pairs])
    else:
        Q = he_all.new_zeros(len(parts), k_max, d)
        P = he_all.new_zeros(len(parts), k_max, d)
        for s, (part, (q, p)) in enumerate(zip(parts, pairs)):
            Q[s, :part.rows_cpu.numel()] = q
            P[s, :part.rows_cpu.numel()] = p
    counts = h2d(torch.tensor([p.rows_cpu.numel() for p in parts], dtype=torch.int32), dev)
    per_problem = criterion.batched(Q, P, counts, symmetric=symmetric)
    return {part.column: per_problem[s] for s, part in enumerate(parts)}


def _local_terms_batched(parts, token_embs, dev, rows_of, subsample=256):
    """{stain column: GOT(he_tokens, stain_tokens, subsample=256)} for every participating stain, the stains' chains running concurrently
    (one autograd node, one HIP stream per stain).  None when there is nothing to overlap or the tokens are not on a ROCm device."""
    if len(parts) < 2 or dev.type != "cuda":
        return None
    from .distributed import got_multi
    # every part's shape is checked BEFORE the first torch.randperm is drawn: the per-stain fallback draws its own, and a batched attempt
    # abandoned half-way would have consumed draws the reference's loop never makes
    if any(token_embs[part.name].squeeze().dim() != 3 for part in parts):
        return None
    problems = []
    for part in parts:
        he_src, st_src = token_embs["HE"][:, :, :, part.column], token_embs[part.name].squeeze()   # .squeeze() as in trainer.py:43
        kk = min(int(part.rows_cpu.numel()), he_src.shape[1])  # randperm(k)[:256] < k: the first k tokens are all GOT can read
        he_tok, st_tok = _take_rows(he_src[:, :kk], part, rows_of), _take_rows(st_src[:, :kk], part, rows_of)
        idx = h2d(torch.randperm(he_tok.shape[0])[:subsample], dev)   # loss.py:282 -- the same draw, in the same order, as GOT()
        problems.append((he_tok.index_select(1, idx).float().contiguous(), st_tok.index_select(1, idx).float().contiguous()))
    outs = got_multi(problems, local=True)
    return {part.column: outs[i, 1] + outs[i, 0] for i, part in enumerate(parts)}


def calculate_losses(STAINS, loss_fn_interMod, loss_fn_interMod_local, loss_fn_intraMod, wsi_embs, token_embs,
                     modality_labels_withoutHE, args):
    """Sum of the active loss terms over the participating stains; returns (loss, at_least_one_stain_flag).
    With madeleine_amd.GOT as the local loss and two or more participating stains, the stains' GOT terms run as ONE batched launch
    sequence (distributed.got_multi(local=True)): every problem then runs in the kernels of the size class of the LARGEST n of the
    batch, so a term can differ from the per-stain GOT() call in fp32 rounding (same arithmetic, other association: <= 1e-5 relative,
    tests/test_bench_path_gpu.py::test_got_multi_c4_rank_shape_vs_fp64_oracle); torch.randperm is consumed in the same stain order."""
    parts = _participants(STAINS, modality_labels_withoutHE.detach().cpu())
    if not parts:
        return -1, False   # trainer.py:72-75: nothing but H&E in this batch

    if loss_fn_interMod and args.global_loss != "info-nce":
        raise AssertionError("invalid global loss")   # the reference asserts the same (trainer.py:36)
    dev = wsi_embs["HE"].device
    rows_of = _RowIndex(dev)
    precomputed: Optional[Dict[int, torch.Tensor]] = None
    if isinstance(loss_fn_interMod, _HipInfoNCE) and loss_fn_interMod.reduction == 'mean':
        precomputed = _global_terms_batched(loss_fn_interMod, parts, wsi_embs, args.symmetric_cl, rows_of)

    # local terms of all stains as ONE node of concurrent GOT chains (distributed.got_multi(local=True): same arithmetic as the
    # per-stain calls below -- thresholds from each problem's own cost matrices, torch.randperm consumed in the same stain order)
    got_terms = _local_terms_batched(parts, token_embs, dev, rows_of) if loss_fn_interMod_local is _GOT else None
    terms = []
    for part in parts:
        rows = rows_of
        if loss_fn_interMod:                                     # global: slide-level InfoNCE, whole-bag view
            if precomputed is not None:
                terms.append(precomputed[part.column])
            else:
                he, st = _pair(wsi_embs, part, rows, WHOLE_VIEW_POSITION)
                terms.append(loss_fn_interMod(query=he, positive_key=st, symmetric=args.symmetric_cl))
        if loss_fn_interMod_local and got_terms is not None:
            terms.append(got_terms[part.column] * args.local_loss_weight)
        elif loss_fn_interMod_local:                             # local: token-level GOT, 256 sub-sampled tokens
            he_src, st_src = token_embs["HE"][:, :, :, part.column], token_embs[part.name].squeeze()   # .squeeze() as in trainer.py:43
            if loss_fn_interMod_local is _GOT and st_src.dim() == 3:
                # our GOT reads token indices randperm(k)[:256] < k = the number of participating cases (the reference's quirk,
                # loss.py:282): narrowing to the first k tokens BEFORE the row gather is exact and keeps the gathers and their
                # backward at [B, k, 128] instead of [B, N, 128]
                kk = min(int(part.rows_cpu.numel()), he_src.shape[1])
                he_src, st_src = he_src[:, :kk], st_src[:, :kk]
            he_tok, st_tok = _take_rows(he_src, part, rows), _take_rows(st_src, part, rows)
            terms.append(loss_fn_interMod_local(he_tok, st_tok, subsample=256) * args.local_loss_weight)
        if loss_fn_intraMod:                                     # intra: the two half-bag views of each modality
            he1, st1 = _pair(wsi_embs, part, rows, 1)
            he2, st2 = _pair(wsi_embs, part, rows, 2)
            terms.append(loss_fn_intraMod(query=he1, positive_key=he2, symmetric=args.symmetric_cl))
            terms.append(loss_fn_intraMod(query=st1, positive_key=st2, symmetric=args.symmetric_cl))
    # stains participate but every loss callable is switched off: the reference trips its consistency assert (trainer.py:75)
    assert terms, "Loss should be -1 if there are no losses to calculate"
    return sum(terms), True


def train_loop(args, loss_fn_interMod, loss_fn_interMod_local, loss_fn_intraMod, ssl_model, epoch, dataloader, optimizer,
               scheduler_warmup, scheduler):
    """One epoch (trainer.py:80-144): returns (summed batch losses, smooth rank of the epoch's H&E slide embeddings)."""
    n_views = 3 if loss_fn_intraMod else 1        # the intra loss needs the two extra half-bag views
    precision = set_model_precision(args.precision)
    use_autocast = precision in (torch.bfloat16, torch.float16)   # for fp32/fp64 torch disables autocast anyway
    in_warmup = epoch <= args.warmup_epochs       # note `<=`, as the reference (trainer.py:128)
    ssl_model.train()

    epoch_loss, busy_seconds, he_embeddings = 0.0, 0.0, []
    for batch_no, data in enumerate(dataloader):
        if epoch == 0 and batch_no == 0:
            print("Using precision:", precision)
        tick = time.time()
        present = data['modality_labels'][:, HE_POSITION + 1:]

        optimizer.zero_grad()
        with torch.amp.autocast(device_type="cuda", dtype=precision, enabled=use_autocast):
            wsi_embs, token_embs = ssl_model(data, device=DEVICE, n_views=n_views)
            loss, has_pairs = calculate_losses(args.STAINS, loss_fn_interMod, loss_fn_interMod_local, loss_fn_intraMod,
                                               wsi_embs, token_embs, present, args)
        he_embeddings.extend(wsi_embs['HE'][:, WHOLE_VIEW_POSITION, :, 0].detach().float().cpu().numpy())
        if not has_pairs:
            print("Skipping batch with only HE")
            continue

        loss.backward()
        optimizer.step()
        (scheduler_warmup if in_warmup else scheduler).step()

        if batch_no % 3 == 0:
            print(f"Loss for batch: {batch_no} = {loss:.3f}")
        epoch_loss += loss.item()
        busy_seconds += time.time() - tick

    rank = smooth_rank_measure(torch.Tensor(np.array(he_embeddings)))
    return epoch_loss, rank
