#!/usr/bin/env python3
"""Generate tests/golden/*.npz by IMPORTING THE REFERENCE (read-only at /root/reference).

Runs only in the build container (the reference never travels to the GPU box, in any form).
What is committed: this script + the KB-sized outputs.  Inputs/weights are regenerated from
oracle/recipe.py on both sides, so a fixture holds expected OUTPUTS (and a few index arrays).

Process-local shims (they exist only inside this generator):
  * empty `wandb` / `h5py` modules so that madeleine.utils.trainer imports,
  * torch.Tensor.cuda -> identity, because madeleine.utils.loss hard-codes .cuda() and this box has no GPU,
  * nn.Dropout.forward optionally replaced by a mask-injecting version to pin train-mode semantics.

Usage:  python oracle/gen_golden.py            (writes tests/golden/)
"""
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("MADELEINE_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden")

sys.path.insert(0, ROOT)
sys.path.insert(0, REF)
for _m in ("wandb", "h5py"):
    sys.modules.setdefault(_m, types.ModuleType(_m))
torch.Tensor.cuda = lambda self, *a, **k: self  # noqa: E731  (CPU-only box)

from oracle import recipe  # noqa: E402
from madeleine.models.Model import MADELEINE  # noqa: E402  (reference)
from madeleine.utils import loss as ref_loss  # noqa: E402  (reference)
from madeleine.utils import trainer as ref_trainer  # noqa: E402  (reference)

torch.set_num_threads(8)
MODS5 = ["HE", "HER2", "PGR", "KI67", "ER"]


def cfg(mods, d_in, act="softmax"):
    return SimpleNamespace(MODALITIES=list(mods), wsi_encoder="abmil", patch_embedding_dim=d_in,
                           wsi_encoder_hidden_dim=512, activation=act, n_heads=4)


def build(mods, d_in, stain_encoding=False, tag="w", act="softmax"):
    m = MADELEINE(cfg(mods, d_in, act), stain_encoding=stain_encoding)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = {k: torch.from_numpy(v) for k, v in recipe.state_dict_recipe(shapes, tag).items()}
    m.load_state_dict(sd, strict=True)
    return m, shapes


def t(shape, key, lo=-1.0, hi=1.0):
    return torch.from_numpy(recipe.uniform(shape, key, lo, hi))


def npy(x):
    return x.detach().cpu().numpy().astype(np.float32) if torch.is_tensor(x) else np.asarray(x)


def grad_summary(model):
    """Per-parameter gradient L2 norm + the first 16 flat entries (enough to pin sign/scale/layout)."""
    out = {}
    for k, p in model.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        out[f"gnorm/{k}"] = npy(g.norm())
        out[f"ghead/{k}"] = npy(g.flatten()[:16])
    return out


# ------------------------------------------------------------------ encoder (eval mode, grads on)
def gen_encoder():
    B, M, N, D = 2, 3, 40, 64
    mods = MODS5[:M]
    model, shapes = build(mods, D)
    model.eval()
    feats = t((B, M, N, D), "enc:feats")
    embs, toks = model({"feats": feats}, device="cpu", train=True, n_views=1)
    out = {"shape": np.array([B, M, N, D])}
    for k in mods:
        out[f"emb/{k}"] = npy(embs[k])
        out[f"tok/{k}"] = npy(toks[k])
    # the embedder's own outputs (A1-A3)
    slide, raw = model.wsi_embedders(feats.view(B * M, N, D), return_attention=True)
    out["slide"] = npy(slide)
    out["raw"] = npy(raw)
    slide2, tokens = model.wsi_embedders(feats.view(B * M, N, D), return_preattn_feats=True)
    out["tokens_head"] = npy(tokens[:, :2])          # [BM,2,512,H]: pins the head interleave
    # one scalar objective -> parameter gradients
    w_e = t((B, 1, 512), "enc:w_e")
    w_t = t((B, N, 128), "enc:w_t")
    obj = sum((embs[k] * (w_e if k != "HE" else w_e.unsqueeze(3))).sum() for k in mods) + \
        sum((toks[k] * (w_t if k != "HE" else w_t.unsqueeze(3))).sum() for k in mods) * 0.01
    model.zero_grad()
    obj.backward()
    out["obj"] = npy(obj)
    out.update(grad_summary(model))
    # other public branches
    out["encode_he"] = npy(model.encode_he(feats[:, 0], "cpu"))
    ev = model({"feats": feats[:, :1]}, device="cpu", train=False)
    out["eval/HE"] = npy(ev["HE"])
    he, raw_att = model({"feats": feats[:, :1]}, device="cpu", train=False, return_attention=True)
    out["att/HE"] = npy(he)
    out["att/raw"] = npy(raw_att)
    # n_views = 3 (numpy RNG split, Model.py:426-429)
    np.random.seed(7)
    e3, t3 = model({"feats": feats}, device="cpu", train=True, n_views=3)
    for k in mods:
        out[f"emb3/{k}"] = npy(e3[k])
    # other activations of BatchedABMIL (abmil.py:56-61)
    for act in ("relu", "leaky_relu", "sigmoid"):
        m2, _ = build(mods, D, act=act)
        m2.eval()
        s2 = m2.wsi_embedders(feats.view(B * M, N, D))
        out[f"slide_act/{act}"] = npy(s2)
    np.savez_compressed(os.path.join(OUT, "encoder.npz"), **out)


# ------------------------------------------------------------------ stain encoding (pins the r//B quirk)
def gen_stain_encoding():
    B, M, N, D = 3, 4, 24, 64
    mods = MODS5[:M]
    model, _ = build(mods, D, stain_encoding=True, tag="wse")
    model.eval()
    feats = t((B, M, N, D), "se:feats")
    embs, toks = model({"feats": feats}, device="cpu", train=True)
    out = {"shape": np.array([B, M, N, D])}
    for k in mods:
        out[f"emb/{k}"] = npy(embs[k])
        out[f"tok_head/{k}"] = npy(toks[k][:, :3])
    # eval branch with stain encoding only works at batch size 1 in the reference
    # (indicator is [1,bs] -> repeat_interleave over dim 1 -> cat fails for bs>1, Model.py:187-189)
    ev = model({"feats": feats[:1, :1]}, device="cpu", train=False)                        # key 0
    out["eval/HE"] = npy(ev["HE"])
    ev2 = model({"feats": feats[:1, 2:3]}, device="cpu", train=False, custom_stain_idx=2)  # key 2
    out["eval/custom2"] = npy(ev2[mods[2]])
    np.savez_compressed(os.path.join(OUT, "stain_encoding.npz"), **out)


# ------------------------------------------------------------------ train mode with injected dropout masks
def gen_train_dropout():
    B, M, N, D = 2, 2, 16, 64
    mods = MODS5[:M]
    model, _ = build(mods, D, tag="wdo")
    model.train()
    feats = t((B, M, N, D), "do:feats")
    BM = B * M
    # call order of nn.Dropout in the reference forward: pre_attn 0,1,2 then per head a, b
    keys = [("do:pre0", (BM, N, 512), 0.1), ("do:pre1", (BM, N, 512), 0.1), ("do:pre2", (BM, N, 2048), 0.1)]
    for c in range(4):
        keys += [(f"do:gate{c}a", (BM, N, 512), 0.25), (f"do:gate{c}b", (BM, N, 512), 0.25)]
    queue = [(torch.from_numpy(recipe.bernoulli(s, k, 1.0 - p)), p) for k, s, p in keys]
    pos = [0]
    orig = torch.nn.Dropout.forward

    def injected(self, x):
        keep, p = queue[pos[0]]
        pos[0] += 1
        assert abs(p - self.p) < 1e-9 and keep.shape == x.shape, (p, self.p, keep.shape, x.shape)
        return x * keep / (1.0 - p)

    torch.nn.Dropout.forward = injected
    try:
        embs, toks = model({"feats": feats}, device="cpu", train=True)
        assert pos[0] == len(queue)
    finally:
        torch.nn.Dropout.forward = orig
    out = {"shape": np.array([B, M, N, D])}
    for k in mods:
        out[f"emb/{k}"] = npy(embs[k])
        out[f"tok_head/{k}"] = npy(toks[k][:, :3])
    crit = ref_loss.InfoNCE(temperature=0.1)
    loss = crit(embs["HE"][:, 0, :, 0], embs[mods[1]][:, 0, :], symmetric=True)
    model.zero_grad()
    loss.backward()
    out["loss"] = npy(loss)
    out.update(grad_summary(model))
    np.savez_compressed(os.path.join(OUT, "train_dropout.npz"), **out)


# ------------------------------------------------------------------ InfoNCE
def gen_infonce():
    out = {}
    for k in (2, 7, 33):
        q0, p0 = t((k, 512), f"nce:q{k}"), t((k, 512), f"nce:p{k}")
        p0 = p0 + 0.1 * q0  # weakly correlated positives: keeps the T=0.001 loss away from exactly 0
        for T in (0.001, 0.1):
            for sym in (False, True):
                q, p = q0.clone().requires_grad_(), p0.clone().requires_grad_()
                loss = ref_loss.InfoNCE(temperature=T)(q, p, symmetric=sym)
                loss.backward()
                tag = f"k{k}/T{T}/sym{int(sym)}"
                out[f"{tag}/loss"] = npy(loss)
                out[f"{tag}/dq_norm"], out[f"{tag}/dp_norm"] = npy(q.grad.norm()), npy(p.grad.norm())
                out[f"{tag}/dq"] = npy(q.grad if k <= 7 else q.grad[:, :32])
                out[f"{tag}/dp"] = npy(p.grad if k <= 7 else p.grad[:, :32])
    np.savez_compressed(os.path.join(OUT, "infonce.npz"), **out)


# ------------------------------------------------------------------ GOT and its pieces
def gen_got():
    out = {}
    N = 40
    for k in (2, 7, 32):
        # The GW fixed-point iteration (5 x 20 IPOT steps at beta=0.1) is chaotic for some instances: the
        # reference itself then moves by 1e-2 relative under a different token permutation (= summation
        # order).  Golden instances are the first recipe key on which the reference agrees with itself
        # to 2e-6 under two permutations; the trial index is stored in the fixture.
        for trial in range(64):
            v0, q0 = t((k, N, 128), f"got:v{k}:{trial}"), t((k, N, 128), f"got:q{k}:{trial}")
            q0 = q0 + 0.7 * v0
            with torch.no_grad():
                torch.manual_seed(1)
                l1 = ref_loss.GOT(v0, q0, subsample=256)
                torch.manual_seed(2)
                l2 = ref_loss.GOT(v0, q0, subsample=256)
            if abs(float(l1) - float(l2)) <= 2e-6 * abs(float(l1)):
                break
        else:
            raise RuntimeError("no well-conditioned GOT instance found")
        out[f"k{k}/trial"] = np.array(trial)
        v, q = v0.clone().requires_grad_(), q0.clone().requires_grad_()
        torch.manual_seed(100 + k)
        perm = torch.randperm(k)              # same draw the reference makes first inside GOT
        torch.manual_seed(100 + k)
        loss = ref_loss.GOT(v, q, subsample=256)
        loss.backward()
        out[f"k{k}/perm"] = perm.numpy()
        out[f"k{k}/loss"] = npy(loss)
        out[f"k{k}/dv_norm"] = npy(v.grad.norm())
        out[f"k{k}/dq_norm"] = npy(q.grad.norm())
        out[f"k{k}/dv_tail_abs"] = npy(v.grad[:, k:].abs().max()) if N > k else np.float32(0)
        if k <= 7:
            out[f"k{k}/dv"] = npy(v.grad[:, :k])
            out[f"k{k}/dq"] = npy(q.grad[:, :k])
        else:
            out[f"k{k}/dv"] = npy(v.grad[:4, :k, :16])
            out[f"k{k}/dq"] = npy(q.grad[:4, :k, :16])
    # pieces on a fixed small problem: pins G1/G2/G3 separately
    k, n = 3, 9
    for trial in range(64):
        v = t((k, n, 128), f"got:pv:{trial}")
        q = t((k, n, 128), f"got:pq:{trial}") + 0.5 * v
        pr = torch.randperm(n, generator=torch.Generator().manual_seed(3))
        with torch.no_grad():
            l1, l2 = ref_loss.GOT(v, q), ref_loss.GOT(v[:, pr], q[:, pr])
        if abs(float(l1) - float(l2)) <= 2e-6 * abs(float(l1)):
            break
    else:
        raise RuntimeError("no well-conditioned GOT piece instance found")
    out["piece/trial"] = np.array(trial)
    x, y = v.transpose(2, 1), q.transpose(2, 1)
    c = ref_loss.cost_matrix_batch_torch(x, y).transpose(1, 2)
    out["piece/cross_cost"] = npy(c)
    out["piece/intra_cost"] = npy(ref_loss.cos_batch_torch(x, x))
    lo, hi = c.min(), c.max()
    cthr = torch.relu(c - (lo + 0.1 * (hi - lo)))
    out["piece/ipot30"] = npy(ref_loss.IPOT_torch_batch_uniform(cthr, k, n, n, beta=0.5, iteration=30))
    out["piece/wd"] = npy(-ref_loss.IPOT_distance_torch_batch_uniform(cthr, k, n, n, 30))
    out["piece/gwd"] = npy(ref_loss.GW_distance_uniform(x, y))
    # no-subsample call (subsample=None): all N tokens
    out["nosub/loss"] = npy(ref_loss.GOT(v, q))
    np.savez_compressed(os.path.join(OUT, "got.npz"), **out)


# ------------------------------------------------------------------ calculate_losses (H1)
def gen_calculate_losses():
    B, M, N = 6, 5, 12
    stains = MODS5[1:]
    wsi, tok = {}, {}
    he_e, he_t = t((B, 1, 512), "cl:he_e"), t((B, N, 128), "cl:he_t")
    wsi["HE"] = he_e.unsqueeze(3).repeat(1, 1, 1, M - 1)
    tok["HE"] = he_t.unsqueeze(3).repeat(1, 1, 1, M - 1)
    for s in stains:
        wsi[s] = t((B, 1, 512), f"cl:e{s}") + 0.1 * he_e
        tok[s] = t((B, N, 128), f"cl:t{s}") + 0.6 * he_t
    labels = torch.tensor([[1, 1, 0, 1, 1],
                           [1, 1, 0, 0, 1],
                           [1, 0, 1, 1, 1],
                           [1, 1, 0, 1, 0],
                           [1, 0, 0, 1, 1],
                           [1, 1, 0, 0, 1]], dtype=torch.float32)  # PGR has k=1 -> skipped
    args = SimpleNamespace(global_loss="info-nce", symmetric_cl=True, local_loss_weight=0.7)
    leaves = {k: v.clone().requires_grad_() for k, v in list(wsi.items()) + [("t" + k, v) for k, v in tok.items()]}
    wsi_l = {k: leaves[k] for k in wsi}
    tok_l = {k: leaves["t" + k] for k in tok}
    out = {"labels": labels.numpy()}
    torch.manual_seed(5)
    loss, flag = ref_trainer.calculate_losses(stains, ref_loss.InfoNCE(temperature=0.001), ref_loss.GOT, None,
                                               wsi_l, tok_l, labels[:, 1:], args)
    loss.backward()
    out["full/loss"], out["full/flag"] = npy(loss), np.array(flag)
    for k in wsi:
        gn = lambda x: np.float32(0) if x.grad is None else npy(x.grad.norm())  # noqa: E731 (skipped stain)
        out[f"full/dwsi_norm/{k}"] = gn(leaves[k])
        out[f"full/dtok_norm/{k}"] = gn(leaves["t" + k])
    # global only
    loss_g, flag_g = ref_trainer.calculate_losses(stains, ref_loss.InfoNCE(temperature=0.001), None, None,
                                                   wsi, tok, labels[:, 1:], args)
    out["global/loss"], out["global/flag"] = npy(loss_g), np.array(flag_g)
    # sentinel: H&E only
    l0 = torch.zeros(B, M)
    l0[:, 0] = 1
    l0[2, 3] = 1
    loss_s, flag_s = ref_trainer.calculate_losses(stains, ref_loss.InfoNCE(temperature=0.001), ref_loss.GOT, None,
                                                   wsi, tok, l0[:, 1:], args)
    out["sentinel/loss"], out["sentinel/flag"] = np.float32(loss_s), np.array(flag_s)
    # intra-modality term (3 views)
    wsi3 = {k: torch.cat([v, t(v.shape, f"cl:v1{k}"), t(v.shape, f"cl:v2{k}")], dim=1) for k, v in wsi.items()}
    loss_i, _ = ref_trainer.calculate_losses(stains, ref_loss.InfoNCE(temperature=0.001), None,
                                             ref_loss.InfoNCE(temperature=0.001), wsi3, tok, labels[:, 1:], args)
    out["intra/loss"] = npy(loss_i)
    np.savez_compressed(os.path.join(OUT, "calculate_losses.npz"), **out)


# ------------------------------------------------------------------ one full fwd + losses + bwd (C1-like, tiny)
def gen_full_step():
    B, M, N, D = 4, 3, 32, 64
    mods = MODS5[:M]
    model, _ = build(mods, D, tag="wfs")
    model.eval()
    feats = t((B, M, N, D), "fs:feats")
    labels = torch.tensor([[1, 1, 1], [1, 1, 0], [1, 1, 1], [1, 0, 1]], dtype=torch.float32)
    args = SimpleNamespace(global_loss="info-nce", symmetric_cl=True, local_loss_weight=1.0)
    embs, toks = model({"feats": feats}, device="cpu", train=True)
    torch.manual_seed(11)
    loss, flag = ref_trainer.calculate_losses(mods[1:], ref_loss.InfoNCE(temperature=0.001), ref_loss.GOT, None,
                                               embs, toks, labels[:, 1:], args)
    model.zero_grad()
    loss.backward()
    out = {"shape": np.array([B, M, N, D]), "labels": labels.numpy(), "loss": npy(loss), "flag": np.array(flag)}
    out.update(grad_summary(model))
    # global-only variant (config-1 plumbing: ABMIL + global InfoNCE)
    embs, toks = model({"feats": feats}, device="cpu", train=True)
    loss_g, _ = ref_trainer.calculate_losses(mods[1:], ref_loss.InfoNCE(temperature=0.001), None, None,
                                             embs, toks, labels[:, 1:], args)
    model.zero_grad()
    loss_g.backward()
    out["global/loss"] = npy(loss_g)
    out.update({"global/" + k: v for k, v in grad_summary(model).items()})
    np.savez_compressed(os.path.join(OUT, "full_step.npz"), **out)


# ------------------------------------------------------------------ H2: a train_loop trajectory (trainer.py:80-144)
SGD_LR = 30.0   # gradients are O(1e-4) at this scale: a large rate makes the three steps visible in the losses
TRAIN_LOOP_LABELS = [[[1, 1, 1], [1, 1, 0], [1, 1, 1], [1, 0, 1]],
                     [[1, 0, 0], [1, 0, 0], [1, 0, 0], [1, 0, 0]],      # H&E only: the loop must skip this batch
                     [[1, 1, 1], [1, 1, 1], [1, 0, 1], [1, 1, 1]],
                     [[1, 1, 0], [1, 1, 1], [1, 1, 1], [1, 1, 1]]]


def train_loop_batches(B, M, N, D):
    return [{"feats": t((B, M, N, D), f"tl:feats{i}"), "modality_labels": torch.tensor(lab, dtype=torch.float32),
             "slide_ids": [f"s{i}_{j}" for j in range(B)]} for i, lab in enumerate(TRAIN_LOOP_LABELS)]


def gen_train_loop():
    """Three optimiser steps (one of four batches is H&E-only and skipped) of the reference's train_loop in train mode.
    The trajectory is pinned by making every nn.Dropout the identity (p = 0 on the modules; train mode stays on) and,
    for the parameter trajectory, by a plain SGD optimiser: AdamW's first steps are sign(g)-like, so an element whose
    gradient is rounding noise moves by +-lr on either side and no implementation can reproduce it -- with AdamW only
    the per-step losses are recorded."""
    B, M, N, D = 4, 3, 40, 64
    mods = MODS5[:M]
    args = SimpleNamespace(STAINS=mods[1:], precision="float32", warmup_epochs=1, global_loss="info-nce", symmetric_cl=True,
                           local_loss_weight=0.5)
    out = {"shape": np.array([B, M, N, D])}
    orig_cl = ref_trainer.calculate_losses
    for opt_name in ("sgd", "adamw"):
        model, _ = build(mods, D, tag="wtl")
        for m in model.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        opt = torch.optim.SGD(model.parameters(), lr=SGD_LR) if opt_name == "sgd" else torch.optim.AdamW(model.parameters(), lr=1e-3)
        warm = torch.optim.lr_scheduler.LinearLR(opt, start_factor=1e-5, total_iters=4)
        cos = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=10, eta_min=1e-8)
        step_losses = []

        def logged(*a, **k):
            loss, flag = orig_cl(*a, **k)
            if flag:
                step_losses.append(float(loss))
            return loss, flag
        ref_trainer.calculate_losses = logged
        try:
            torch.manual_seed(0)
            ep_loss, rank = ref_trainer.train_loop(args, ref_loss.InfoNCE(temperature=0.1), ref_loss.GOT, None, model, 5,
                                                   train_loop_batches(B, M, N, D), opt, warm, cos)
        finally:
            ref_trainer.calculate_losses = orig_cl
        assert len(step_losses) == 3 and cos.last_epoch == 3 and warm.last_epoch == 0
        out[f"{opt_name}/step_losses"] = np.array(step_losses, dtype=np.float64)
        out[f"{opt_name}/ep_loss"] = np.float64(ep_loss)
        out[f"{opt_name}/rank"] = np.float64(rank)
        if opt_name == "sgd":
            for k, v in model.state_dict().items():
                out[f"sgd/pnorm/{k}"] = npy(v.norm())
                out[f"sgd/phead/{k}"] = npy(v.flatten()[:16])
            ref0 = {k: torch.from_numpy(v) for k, v in recipe.state_dict_recipe({k: tuple(v.shape) for k, v in model.state_dict().items()}, "wtl").items()}
            for k, v in model.state_dict().items():
                out[f"sgd/dnorm/{k}"] = npy((v - ref0[k]).norm())        # how far the three steps moved the tensor
                out[f"sgd/dhead/{k}"] = npy((v - ref0[k]).flatten()[:16])
    np.savez_compressed(os.path.join(OUT, "train_loop.npz"), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    only = sys.argv[1:]
    for fn in (gen_encoder, gen_stain_encoding, gen_train_dropout, gen_infonce, gen_got,
               gen_calculate_losses, gen_full_step, gen_train_loop):
        if only and fn.__name__ not in only:
            continue
        fn()
        print("ok", fn.__name__)
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print(f"fixtures: {tot / 1024:.0f} KiB in {OUT}")


if __name__ == "__main__":
    main()
