"""CPU restatement (plain PyTorch fp32, autograd) of the MADELEINE pretrain hot path.

TEST INFRASTRUCTURE ONLY.  This file is the *checker*: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import it.  The product (madeleine_amd/) never does and
fails loudly when its HIP extension is missing.

Parity pin: every function below is checked against golden vectors produced by importing the
reference itself in the build container (oracle/gen_golden.py -> tests/golden/*.npz,
tests/test_oracle_golden.py).  The reference ships no tests of its own,
so those fixtures are the only pin.

All `file:line` citations are relative to the reference checkout (/root/reference), which
never travels to the GPU box.  The code is written functionally over a flat parameter dict
that uses the reference's state_dict key names.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

PRE_DROPOUT_P = 0.1    # Model.py:354,358,362
GATE_DROPOUT_P = 0.25  # abmil.py:33-35
HIDDEN = 512           # attention hidden dim is hard-wired, Model.py:71
TOKEN_DIM = 128        # Model.py:80-83


# --------------------------------------------------------------------------------------
# A1: pre-attention MLP   (Model.py:346-363, called at :393)
# --------------------------------------------------------------------------------------
def _drop(x: torch.Tensor, keep: Optional[torch.Tensor], p: float) -> torch.Tensor:
    """Inverted dropout with an injected keep-mask (1 = keep); identity when keep is None (.eval())."""
    if keep is None:
        return x
    return x * keep * (1.0 / (1.0 - p))


def pre_attn(bags: torch.Tensor, sd: Params, prefix: str = "wsi_embedders.pre_attn.",
             keep_masks: Optional[Sequence[torch.Tensor]] = None) -> torch.Tensor:
    """3 x (Linear -> LayerNorm(eps 1e-5) -> GELU(erf) -> Dropout .1): D -> 512 -> 512 -> 512*H."""
    x = bags
    for blk, (lin, ln) in enumerate(((0, 1), (4, 5), (8, 9))):
        w, b = sd[f"{prefix}{lin}.weight"], sd[f"{prefix}{lin}.bias"]
        g, beta = sd[f"{prefix}{ln}.weight"], sd[f"{prefix}{ln}.bias"]
        x = F.linear(x, w, b)
        x = F.layer_norm(x, (w.shape[0],), g, beta, eps=1e-5)
        x = F.gelu(x)  # exact erf form
        x = _drop(x, None if keep_masks is None else keep_masks[blk], PRE_DROPOUT_P)
    return x


# --------------------------------------------------------------------------------------
# A2: gated attention score of one head   (abmil.py:41-68)
# --------------------------------------------------------------------------------------
def gate_scores(x: torch.Tensor, wa, ba, wb, bb, wc, bc,
                keep_a: Optional[torch.Tensor] = None, keep_b: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [BM,N,512] -> raw score [BM,N,1]:  Wc (drop(tanh(Wa x)) * drop(sigmoid(Wb x))) + bc."""
    a = _drop(torch.tanh(F.linear(x, wa, ba)), keep_a, GATE_DROPOUT_P)
    b = _drop(torch.sigmoid(F.linear(x, wb, bb)), keep_b, GATE_DROPOUT_P)
    return F.linear(a * b, wc, bc)


def activate(raw: torch.Tensor, activation: str) -> torch.Tensor:
    """abmil.py:54-63; softmax is over the PATCH axis (dim=1)."""
    if activation == "softmax":
        return torch.softmax(raw, dim=1)
    if activation == "leaky_relu":
        return F.leaky_relu(raw)
    if activation == "relu":
        return F.relu(raw)
    if activation == "sigmoid":
        return torch.sigmoid(raw)
    raise NotImplementedError("Activation not implemented.")


# --------------------------------------------------------------------------------------
# A1-A3: ABMILEmbedder.forward   (Model.py:375-451)
# --------------------------------------------------------------------------------------
def abmil_embed(bags: torch.Tensor, sd: Params, n_heads: int = 4, activation: str = "softmax",
                pre_keep=None, gate_keep=None, view_indices: Optional[List[torch.Tensor]] = None):
    """bags [BM,N,Din] -> dict(slide [BM,(V,)512,H], tokens [BM,N,512,H], raw [BM,N,1,H]).

    Head c owns channels j = e*H + c of the 2048-vector (rearrange 'b t (e c) -> b t e c', Model.py:396).
    gate_keep: optional list over heads of (keep_a, keep_b) masks [BM,N,512].
    view_indices: optional [idx_view1, idx_view2] token index sets; reproduces the n_views=3 branch
    (Model.py:419-440) with the split supplied by the caller instead of np.random.shuffle.
    """
    e = pre_attn(bags, sd, keep_masks=pre_keep)
    bm, n, _ = e.shape
    e = e.view(bm, n, HIDDEN, n_heads) if n_heads > 1 else e.unsqueeze(-1)
    raws, acts = [], []
    for c in range(n_heads):
        p = f"wsi_embedders.attn.{c}."
        ka, kb = (None, None) if gate_keep is None else gate_keep[c]
        raw = gate_scores(e[:, :, :, c], sd[p + "attention_a.0.weight"], sd[p + "attention_a.0.bias"],
                          sd[p + "attention_b.0.weight"], sd[p + "attention_b.0.bias"],
                          sd[p + "attention_c.weight"], sd[p + "attention_c.bias"], ka, kb)
        raws.append(raw)
        acts.append(activate(raw, activation))
    raw = torch.stack(raws, dim=-1)       # [BM,N,1,H]
    att = torch.stack(acts, dim=-1)       # [BM,N,1,H]
    slide = (e * att).sum(dim=1)          # [BM,512,H]   Model.py:416-417
    if view_indices is not None:
        views = [slide.unsqueeze(1)]
        for idx in view_indices:          # re-softmax of the RAW scores over each subset, Model.py:435
            a_v = torch.softmax(raw[:, idx], dim=1)
            views.append((e[:, idx] * a_v).sum(dim=1).unsqueeze(1))
        slide = torch.cat(views, dim=1)   # [BM,3,512,H]
    return {"slide": slide, "tokens": e, "raw": raw}


# --------------------------------------------------------------------------------------
# A0: MADELEINE.forward(train=True)   (Model.py:120-159) and friends
# --------------------------------------------------------------------------------------
def _stain_concat_train(feats: torch.Tensor, sd: Params, bs: int, n_mod: int) -> torch.Tensor:
    """Train-branch stain encoding (Model.py:125-132).  QUIRK reproduced on purpose: the indicator list
    is built stain-major ([0]*bs + [1]*bs + ...) while the flattened rows are case-major, so flattened
    row r receives embedding index r // bs  (SURVEY.md section 8(a) row A0)."""
    idx = torch.arange(bs * n_mod) // bs
    enc = sd["embedding.weight"][idx]                         # [BM,32]
    enc = enc.unsqueeze(1).expand(-1, feats.shape[1], -1)
    return torch.cat([feats, enc], dim=-1)


def madeleine_forward_train(feats: torch.Tensor, sd: Params, modalities: Sequence[str], n_heads: int = 4,
                            activation: str = "softmax", stain_encoding: bool = False,
                            pre_keep=None, gate_keep=None, view_indices=None):
    """feats [B,M,N,D] -> (all_embeddings, all_token_embeddings) with the reference's shapes:
    stain -> [B,V,512] / [B,N,128];  'HE' -> [B,V,512,M-1] / [B,N,128,M-1]."""
    bs, n_mod, n_tok, d_in = feats.shape
    x = feats.reshape(bs * n_mod, n_tok, d_in)
    if stain_encoding:
        x = _stain_concat_train(x, sd, bs, n_mod)
    out = abmil_embed(x, sd, n_heads, activation, pre_keep, gate_keep, view_indices)
    tok = out["tokens"].reshape(bs, n_mod, n_tok, -1)                       # flat 2048, head fastest
    tok = F.linear(tok, sd["token_projector.weight"], sd["token_projector.bias"])  # [B,M,N,128]
    slide = out["slide"]
    slide = slide.reshape(bs * n_mod, -1, HIDDEN * n_heads)                 # [BM,V,2048]
    slide = F.linear(slide, sd["projector.weight"], sd["projector.bias"]).view(bs, n_mod, -1, HIDDEN)
    embs, toks = {}, {}
    for i, name in enumerate(modalities):
        s, t = slide[:, i], tok[:, i]
        if name == "HE":
            s = s.unsqueeze(3).repeat(1, 1, 1, n_mod - 1)
            t = t.unsqueeze(3).repeat(1, 1, 1, n_mod - 1)
        embs[name], toks[name] = s, t
    return embs, toks


def madeleine_forward_eval(feats: torch.Tensor, sd: Params, modalities: Sequence[str], n_heads: int = 4,
                           activation: str = "softmax", stain_encoding: bool = False,
                           custom_stain_idx: Optional[int] = None):
    """Eval branch (Model.py:162-203): loops stains, true stain index for the encoding (:186-190).
    As in the reference the final .view(bs, n_mod, d) only works for n_mod == 1."""
    bs, n_mod, n_tok, _ = feats.shape
    out = {}
    for s in range(n_mod):
        name = modalities[custom_stain_idx] if custom_stain_idx else modalities[s]
        x = feats[:, s]
        if stain_encoding:
            key = custom_stain_idx if custom_stain_idx else s
            enc = sd["embedding.weight"][key].view(1, 1, -1).expand(bs, n_tok, -1)
            x = torch.cat([x, enc], dim=-1)
        slide = abmil_embed(x, sd, n_heads, activation)["slide"]            # [bs,512,H]
        slide = slide.reshape(bs * n_mod, HIDDEN * n_heads)
        out[name] = F.linear(slide, sd["projector.weight"], sd["projector.bias"]).view(bs, n_mod, HIDDEN)
    return out


def encode_he(feats: torch.Tensor, sd: Params, n_heads: int = 4, activation: str = "softmax") -> torch.Tensor:
    """Model.py:97-107: [B,N,D] -> [B,512]."""
    slide = abmil_embed(feats, sd, n_heads, activation)["slide"]
    slide = slide.reshape(feats.shape[0], HIDDEN * n_heads)
    return F.linear(slide, sd["projector.weight"], sd["projector.bias"])


# --------------------------------------------------------------------------------------
# L1: InfoNCE   (loss.py:58-133), only the negative_keys=None branch is functional
# --------------------------------------------------------------------------------------
def info_nce(query: torch.Tensor, positive_key: torch.Tensor, temperature: float = 0.1,
             symmetric: bool = False, reduction: str = "mean") -> torch.Tensor:
    if query.dim() != 2:
        raise ValueError("<query> must have 2 dimensions.")
    if positive_key.dim() != 2:
        raise ValueError("<positive_key> must have 2 dimensions.")
    if len(query) != len(positive_key):
        raise ValueError("<query> and <positive_key> must must have the same number of samples.")
    if query.shape[-1] != positive_key.shape[-1]:
        raise ValueError("Vectors of <query> and <positive_key> should have the same number of components.")
    q = F.normalize(query, dim=-1)          # x / max(|x|, 1e-12), loss.py:132
    k = F.normalize(positive_key, dim=-1)
    logits = q @ k.t()
    labels = torch.arange(len(q), device=q.device)
    if symmetric:                           # loss.py:120-123
        return 0.5 * F.cross_entropy(logits / temperature, labels, reduction=reduction) + \
               0.5 * F.cross_entropy(logits.t() / temperature, labels, reduction=reduction)
    return F.cross_entropy(logits / temperature, labels, reduction=reduction)


# --------------------------------------------------------------------------------------
# G1: cosine-distance costs   (loss.py:162-176, 210-233)
# --------------------------------------------------------------------------------------
def _unit_tokens(t: torch.Tensor) -> torch.Tensor:
    """[k,n,d] tokens -> x / (|x|_2 + 1e-12) along d  (eps ADDED to the norm, loss.py:172-173)."""
    return t / (t.norm(p=2, dim=2, keepdim=True) + 1e-12)


def cross_cost(v: torch.Tensor, q: torch.Tensor) -> torch.Tensor:
    """C[b,i,j] = 1 - <v_hat[b,i], q_hat[b,j]>   ([k,n,d],[k,m,d] -> [k,n,m]).
    (cost_matrix_batch_torch returns the transpose, GOT transposes it back: loss.py:176,287)."""
    return 1.0 - torch.bmm(_unit_tokens(v), _unit_tokens(q).transpose(1, 2))


def threshold_relu(c: torch.Tensor, beta: float = 0.1) -> torch.Tensor:
    """thr = min + beta*(max-min) over the WHOLE batch tensor, relu(c - thr)  (loss.py:288-292, 226-233)."""
    lo, hi = c.min(), c.max()
    return torch.relu(c - (lo + beta * (hi - lo)))


def intra_cost(x: torch.Tensor) -> torch.Tensor:
    """cos_batch_torch(X, X) (loss.py:210-233): thresholded self-distance, returned TRANSPOSED (:233).
    1 - <x_i,x_j> is symmetric only up to bmm rounding, so the transpose is kept."""
    xn = _unit_tokens(x)
    c = 1.0 - torch.bmm(xn, xn.transpose(1, 2))
    return threshold_relu(c).transpose(1, 2)


# --------------------------------------------------------------------------------------
# G2: IPOT   (loss.py:179-193)
# --------------------------------------------------------------------------------------
def ipot(c: torch.Tensor, beta: float = 0.5, iteration: int = 50) -> torch.Tensor:
    """c [k,n,m] -> transport plan T [k,n,m]; uniform marginals; autograd flows through every iteration."""
    k, n, m = c.shape
    sigma = torch.full((k, m, 1), 1.0 / m, dtype=c.dtype)
    t = torch.ones(k, n, m, dtype=c.dtype)
    a = torch.exp(-c / beta)
    for _ in range(iteration):
        q = a * t
        delta = 1.0 / (n * torch.bmm(q, sigma))
        sigma = 1.0 / (float(m) * torch.bmm(q.transpose(1, 2), delta))
        t = delta * q * sigma.transpose(1, 2)
    return t


def _trace_ct(c: torch.Tensor, t: torch.Tensor) -> torch.Tensor:
    """trace(C^T T) per batch element = sum_ij C_ij T_ij  (batch_trace of bmm(C^T,T), loss.py:196-206)."""
    return (c * t).sum(dim=(1, 2))


# --------------------------------------------------------------------------------------
# G3: Gromov-Wasserstein   (loss.py:236-275)
# --------------------------------------------------------------------------------------
def gw_distance(x: torch.Tensor, y: torch.Tensor, lam: float = 0.1, iteration: int = 5,
                ot_iteration: int = 20) -> torch.Tensor:
    """x,y [k,n,d] token sets -> [k] GW distances (uniform marginals)."""
    cs, ct = intra_cost(x), intra_cost(y)          # [k,n,n], [k,m,m]
    k, n, m = cs.shape[0], cs.shape[2], ct.shape[2]
    p = torch.full((k, n, 1), 1.0 / n, dtype=x.dtype)   # NB loss.py:270-275 swaps the names m/n; equal here
    q = torch.full((k, m, 1), 1.0 / m, dtype=x.dtype)
    cst = torch.bmm(cs ** 2, p) + torch.bmm(q.transpose(1, 2), (ct ** 2).transpose(1, 2))  # [k,n,1]+[k,1,m]
    gamma = torch.bmm(p, q.transpose(1, 2))
    for _ in range(iteration):
        c_gamma = cst - 2.0 * torch.bmm(torch.bmm(cs, gamma), ct.transpose(1, 2))
        gamma = ipot(c_gamma, beta=lam, iteration=ot_iteration)
    c_gamma = cst - 2.0 * torch.bmm(torch.bmm(cs, gamma), ct.transpose(1, 2))
    return _trace_ct(c_gamma, gamma.detach())       # only the RETURNED gamma is detached (loss.py:248)


# --------------------------------------------------------------------------------------
# G0: GOT   (loss.py:278-302)
# --------------------------------------------------------------------------------------
def got(v_: torch.Tensor, q_: torch.Tensor, subsample: Optional[int] = None,
        perm: Optional[torch.Tensor] = None) -> torch.Tensor:
    """v_,q_ [k,N,128] -> 0-d loss = sum_b GW_b + sum_b WD_b.

    QUIRK reproduced: the sub-sample indices are randperm(v_.shape[0]) -- the masked BATCH size k,
    not N -- so the first n=min(k,subsample) tokens of every bag are used (in a random order which
    does not change the value beyond summation order).  `perm` injects that permutation for tests."""
    if subsample is not None:
        idx = torch.randperm(v_.shape[0]) if perm is None else perm
        idx = idx[:subsample]
        v_, q_ = v_[:, idx, :], q_[:, idx, :]
    c = threshold_relu(cross_cost(v_, q_))
    wd = _trace_ct(c, ipot(c, beta=0.5, iteration=30)).sum()
    gwd = gw_distance(v_, q_).sum()
    return gwd + wd


# --------------------------------------------------------------------------------------
# GOT with externally supplied threshold extrema (the data-parallel decomposition).
# Not a reference function: with extrema=None it IS got() above (asserted in tests/test_oracle_golden.py);
# with the global-batch extrema, per-rank sums add up to the reference's global-batch value.
# --------------------------------------------------------------------------------------
def got_raw_costs(v: torch.Tensor, q: torch.Tensor):
    vn, qn = _unit_tokens(v), _unit_tokens(q)
    return (1.0 - torch.bmm(vn, qn.transpose(1, 2)), 1.0 - torch.bmm(vn, vn.transpose(1, 2)),
            1.0 - torch.bmm(qn, qn.transpose(1, 2)))


def got_extrema(v: torch.Tensor, q: torch.Tensor) -> torch.Tensor:
    c0, cs0, ct0 = got_raw_costs(v, q)
    return torch.stack([c0.min(), c0.max(), cs0.min(), cs0.max(), ct0.min(), ct0.max()])


def got_parts(v: torch.Tensor, q: torch.Tensor, extrema: Optional[torch.Tensor] = None) -> torch.Tensor:
    """-> tensor [2] = (sum_b WD_b, sum_b GWD_b); thresholds from `extrema` [6] if given, else batch-local."""
    c0, cs0, ct0 = got_raw_costs(v, q)
    ex = got_extrema(v, q) if extrema is None else extrema
    thr = lambda m: ex[2 * m] + 0.1 * (ex[2 * m + 1] - ex[2 * m])  # noqa: E731
    c = torch.relu(c0 - thr(0))
    wd = _trace_ct(c, ipot(c, beta=0.5, iteration=30)).sum()
    cs, ct = torch.relu(cs0 - thr(1)).transpose(1, 2), torch.relu(ct0 - thr(2)).transpose(1, 2)
    k, n, m = cs.shape[0], cs.shape[2], ct.shape[2]
    p = torch.full((k, n, 1), 1.0 / n, dtype=v.dtype)
    qq = torch.full((k, m, 1), 1.0 / m, dtype=v.dtype)
    cst = torch.bmm(cs ** 2, p) + torch.bmm(qq.transpose(1, 2), (ct ** 2).transpose(1, 2))
    gamma = torch.bmm(p, qq.transpose(1, 2))
    for _ in range(5):
        gamma = ipot(cst - 2.0 * torch.bmm(torch.bmm(cs, gamma), ct.transpose(1, 2)), beta=0.1, iteration=20)
    c_gamma = cst - 2.0 * torch.bmm(torch.bmm(cs, gamma), ct.transpose(1, 2))
    return torch.stack([wd, _trace_ct(c_gamma, gamma.detach()).sum()])


# --------------------------------------------------------------------------------------
# H1: calculate_losses   (trainer.py:20-77)
# --------------------------------------------------------------------------------------
def calculate_losses(stains, loss_global, loss_local, loss_intra, wsi_embs, token_embs,
                     labels_without_he: torch.Tensor, temperature_symmetric: bool, local_weight: float,
                     global_loss_name: str = "info-nce"):
    """Returns (loss, flag).  loss == -1 and flag False when no stain has >1 cases (trainer.py:71-75)."""
    losses, flag = [], False
    for s_idx, stain in enumerate(stains):
        mask = labels_without_he[:, s_idx].bool()
        if mask.sum().item() > 1:
            if loss_global:
                if global_loss_name != "info-nce":
                    raise AssertionError("invalid global loss")
                he = wsi_embs["HE"][:, 0, :, s_idx][mask]
                st = wsi_embs[stain][:, 0, :][mask]
                losses.append(loss_global(he, st, symmetric=temperature_symmetric))
            if loss_local:
                he_t = token_embs["HE"][:, :, :, s_idx][mask]
                st_t = token_embs[stain].squeeze()[mask]
                losses.append(loss_local(he_t, st_t, subsample=256) * local_weight)
            if loss_intra:
                for emb, sel in ((wsi_embs["HE"], lambda e, v: e[:, v, :, s_idx]),
                                 (wsi_embs[stain], lambda e, v: e[:, v, :])):
                    losses.append(loss_intra(sel(emb, 1)[mask], sel(emb, 2)[mask], symmetric=temperature_symmetric))
            flag = True
    if losses:
        return sum(losses), flag
    return -1, flag


# --------------------------------------------------------------------------------------
# whole pretrain step on the CPU (bench.py cpu_baseline leg and end-to-end parity)
# --------------------------------------------------------------------------------------
def param_shapes(n_mod: int, d_in: int = 512, n_heads: int = 4, stain_encoding: bool = False) -> Dict[str, tuple]:
    """state_dict key -> shape, in the reference's key order (Model.py:46-94)."""
    din = d_in + (32 if stain_encoding else 0)
    sh: Dict[str, tuple] = {}
    if stain_encoding:
        sh["embedding.weight"] = (n_mod, 32)
    sh["token_projector.weight"], sh["token_projector.bias"] = (TOKEN_DIM, HIDDEN * n_heads), (TOKEN_DIM,)
    for i, (o, n_in) in zip((0, 4, 8), ((HIDDEN, din), (HIDDEN, HIDDEN), (HIDDEN * n_heads, HIDDEN))):
        sh[f"wsi_embedders.pre_attn.{i}.weight"], sh[f"wsi_embedders.pre_attn.{i}.bias"] = (o, n_in), (o,)
        sh[f"wsi_embedders.pre_attn.{i + 1}.weight"], sh[f"wsi_embedders.pre_attn.{i + 1}.bias"] = (o,), (o,)
    for c in range(n_heads):
        for g in ("a.0", "b.0"):
            sh[f"wsi_embedders.attn.{c}.attention_{g}.weight"] = (HIDDEN, HIDDEN)
            sh[f"wsi_embedders.attn.{c}.attention_{g}.bias"] = (HIDDEN,)
        sh[f"wsi_embedders.attn.{c}.attention_c.weight"] = (1, HIDDEN)
        sh[f"wsi_embedders.attn.{c}.attention_c.bias"] = (1,)
    sh["projector.weight"], sh["projector.bias"] = (HIDDEN, HIDDEN * n_heads), (HIDDEN,)
    return sh


def make_params(n_mod: int, d_in: int = 512, n_heads: int = 4, stain_encoding: bool = False,
                seed: int = 42) -> Params:
    """Default-torch-init-like parameters under manual_seed (timing only; parity uses oracle.recipe)."""
    g = torch.Generator().manual_seed(seed)
    sd: Params = {}

    def lin(name, out_f, in_f):
        s = 1.0 / math.sqrt(in_f)
        sd[name + ".weight"] = (torch.rand(out_f, in_f, generator=g) * 2 - 1) * s
        sd[name + ".bias"] = (torch.rand(out_f, generator=g) * 2 - 1) * s

    din = d_in + (32 if stain_encoding else 0)
    if stain_encoding:
        sd["embedding.weight"] = torch.randn(n_mod, 32, generator=g)
    lin("token_projector", TOKEN_DIM, HIDDEN * n_heads)
    for i, (o, n_in) in zip((0, 4, 8), ((HIDDEN, din), (HIDDEN, HIDDEN), (HIDDEN * n_heads, HIDDEN))):
        lin(f"wsi_embedders.pre_attn.{i}", o, n_in)
        sd[f"wsi_embedders.pre_attn.{i + 1}.weight"] = torch.ones(o)
        sd[f"wsi_embedders.pre_attn.{i + 1}.bias"] = torch.zeros(o)
    for c in range(n_heads):
        lin(f"wsi_embedders.attn.{c}.attention_a.0", HIDDEN, HIDDEN)
        lin(f"wsi_embedders.attn.{c}.attention_b.0", HIDDEN, HIDDEN)
        lin(f"wsi_embedders.attn.{c}.attention_c", 1, HIDDEN)
    lin("projector", HIDDEN, HIDDEN * n_heads)
    return sd


def random_keep_masks(bm: int, n: int, n_heads: int, gen: torch.Generator):
    """Bernoulli keep-masks for a train-mode step (the reference's nn.Dropout draws)."""
    pre = [(torch.rand(bm, n, w, generator=gen) >= PRE_DROPOUT_P).float()
           for w in (HIDDEN, HIDDEN, HIDDEN * n_heads)]
    gate = [((torch.rand(bm, n, HIDDEN, generator=gen) >= GATE_DROPOUT_P).float(),
             (torch.rand(bm, n, HIDDEN, generator=gen) >= GATE_DROPOUT_P).float()) for _ in range(n_heads)]
    return pre, gate


def pretrain_step_loss(feats: torch.Tensor, labels: torch.Tensor, sd: Params, modalities: Sequence[str],
                       temperature: float = 0.001, symmetric: bool = True, use_got: bool = False,
                       local_weight: float = 1.0, stain_encoding: bool = False, pre_keep=None, gate_keep=None):
    """fwd + losses of one step (zero_grad/backward/AdamW are the caller's): returns (loss, flag, embs)."""
    embs, toks = madeleine_forward_train(feats, sd, modalities, stain_encoding=stain_encoding,
                                         pre_keep=pre_keep, gate_keep=gate_keep)
    g = lambda a, b, symmetric=False: info_nce(a, b, temperature, symmetric)
    loc = (lambda a, b, subsample=None: got(a, b, subsample)) if use_got else None
    loss, flag = calculate_losses(modalities[1:], g, loc, None, embs, toks, labels[:, 1:], symmetric, local_weight)
    return loss, flag, embs
