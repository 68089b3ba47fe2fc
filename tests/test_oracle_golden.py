"""Pins the CPU restatement (oracle/restatement.py) against golden vectors produced by the imported
reference (oracle/gen_golden.py).  CPU only.  Tolerance 1e-5 relative (both sides are fp32 torch on
the same BLAS; differences come from op ordering only)."""
import numpy as np
import pytest
import torch

from oracle import recipe
from oracle import restatement as R
from tests._util import MODS5, golden, max_rel, recipe_params, rel_err, t

TOL = 1e-5


def _grads_match(g, sd, prefix="", tol=2e-5):
    # attention_c.bias has a mathematically zero gradient (softmax is shift invariant): both sides hold
    # rounding noise there, so norms are compared with an absolute floor tied to the largest gradient.
    top = max(float(g[f"{prefix}gnorm/{k}"]) for k in sd)
    for k, p in sd.items():
        ref_n = float(g[f"{prefix}gnorm/{k}"])
        got = p.grad if p.grad is not None else torch.zeros_like(p)
        assert abs(float(got.norm()) - ref_n) <= tol * ref_n + 1e-6 * top, k
        head = torch.from_numpy(g[f"{prefix}ghead/{k}"])
        assert rel_err(got.flatten()[:16], head) < 1e-4 or float(head.norm()) < 1e-5 * top, k


def test_encoder_eval_and_grads():
    g = golden("encoder")
    B, M, N, D = (int(x) for x in g["shape"])
    mods = MODS5[:M]
    sd = recipe_params(M, D, "w", requires_grad=True)
    feats = t((B, M, N, D), "enc:feats")
    embs, toks = R.madeleine_forward_train(feats, sd, mods)
    for k in mods:
        assert embs[k].shape == g[f"emb/{k}"].shape
        assert rel_err(embs[k], g[f"emb/{k}"]) < TOL
        assert rel_err(toks[k], g[f"tok/{k}"]) < TOL
    out = R.abmil_embed(feats.view(B * M, N, D), sd)
    assert rel_err(out["slide"], g["slide"]) < TOL
    assert max_rel(out["raw"], g["raw"]) < 1e-4
    assert rel_err(out["tokens"][:, :2], g["tokens_head"]) < TOL
    w_e, w_t = t((B, 1, 512), "enc:w_e"), t((B, N, 128), "enc:w_t")
    obj = sum((embs[k] * (w_e if k != "HE" else w_e.unsqueeze(3))).sum() for k in mods) + \
        sum((toks[k] * (w_t if k != "HE" else w_t.unsqueeze(3))).sum() for k in mods) * 0.01
    obj.backward()
    assert abs(float(obj) - float(g["obj"])) < 1e-4 * abs(float(g["obj"]))
    _grads_match(g, sd)


def test_encoder_other_branches():
    g = golden("encoder")
    B, M, N, D = (int(x) for x in g["shape"])
    mods = MODS5[:M]
    sd = recipe_params(M, D, "w")
    feats = t((B, M, N, D), "enc:feats")
    assert rel_err(R.encode_he(feats[:, 0], sd), g["encode_he"]) < TOL
    ev = R.madeleine_forward_eval(feats[:, :1], sd, mods)
    assert rel_err(ev["HE"], g["eval/HE"]) < TOL
    assert rel_err(ev["HE"], g["att/HE"]) < TOL            # return_attention branch = same H&E embedding
    raw = R.abmil_embed(feats[:, 0], sd)["raw"]
    assert max_rel(raw, g["att/raw"]) < 1e-4
    # n_views = 3: replay numpy's shuffle (Model.py:426-429)
    np.random.seed(7)
    idx = np.arange(N)
    np.random.shuffle(idx)
    views = [torch.from_numpy(idx[: N // 2].copy()), torch.from_numpy(idx[N // 2:].copy())]
    e3, _ = R.madeleine_forward_train(feats, sd, mods, view_indices=views)
    for k in mods:
        assert e3[k].shape == g[f"emb3/{k}"].shape
        assert rel_err(e3[k], g[f"emb3/{k}"]) < TOL
    for act in ("relu", "leaky_relu", "sigmoid"):
        s = R.abmil_embed(feats.view(B * M, N, D), sd, activation=act)["slide"]
        assert rel_err(s, g[f"slide_act/{act}"]) < TOL
    with pytest.raises(NotImplementedError):
        R.activate(torch.zeros(1, 2, 1), "nope")


def test_stain_encoding_quirk():
    g = golden("stain_encoding")
    B, M, N, D = (int(x) for x in g["shape"])
    mods = MODS5[:M]
    sd = recipe_params(M, D, "wse", stain_encoding=True)
    feats = t((B, M, N, D), "se:feats")
    embs, toks = R.madeleine_forward_train(feats, sd, mods, stain_encoding=True)
    for k in mods:
        assert rel_err(embs[k], g[f"emb/{k}"]) < TOL
        assert rel_err(toks[k][:, :3], g[f"tok_head/{k}"]) < TOL
    ev = R.madeleine_forward_eval(feats[:1, :1], sd, mods, stain_encoding=True)
    assert rel_err(ev["HE"], g["eval/HE"]) < TOL
    ev2 = R.madeleine_forward_eval(feats[:1, 2:3], sd, mods, stain_encoding=True, custom_stain_idx=2)
    assert rel_err(ev2[mods[2]], g["eval/custom2"]) < TOL


def test_train_mode_dropout_masks():
    g = golden("train_dropout")
    B, M, N, D = (int(x) for x in g["shape"])
    mods = MODS5[:M]
    BM = B * M
    sd = recipe_params(M, D, "wdo", requires_grad=True)
    feats = t((B, M, N, D), "do:feats")
    pre = [torch.from_numpy(recipe.bernoulli((BM, N, w), f"do:pre{i}", 0.9)) for i, w in enumerate((512, 512, 2048))]
    gate = [(torch.from_numpy(recipe.bernoulli((BM, N, 512), f"do:gate{c}a", 0.75)),
             torch.from_numpy(recipe.bernoulli((BM, N, 512), f"do:gate{c}b", 0.75))) for c in range(4)]
    embs, toks = R.madeleine_forward_train(feats, sd, mods, pre_keep=pre, gate_keep=gate)
    for k in mods:
        assert rel_err(embs[k], g[f"emb/{k}"]) < TOL
        assert rel_err(toks[k][:, :3], g[f"tok_head/{k}"]) < TOL
    loss = R.info_nce(embs["HE"][:, 0, :, 0], embs[mods[1]][:, 0, :], 0.1, symmetric=True)
    loss.backward()
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    _grads_match(g, sd, tol=1e-4)


@pytest.mark.parametrize("k", [2, 7, 33])
@pytest.mark.parametrize("T", [0.001, 0.1])
@pytest.mark.parametrize("sym", [False, True])
def test_infonce(k, T, sym):
    g = golden("infonce")
    q0, p0 = t((k, 512), f"nce:q{k}"), t((k, 512), f"nce:p{k}")
    p0 = p0 + 0.1 * q0
    q, p = q0.clone().requires_grad_(), p0.clone().requires_grad_()
    loss = R.info_nce(q, p, T, sym)
    loss.backward()
    tag = f"k{k}/T{T}/sym{int(sym)}"
    assert abs(float(loss) - float(g[f"{tag}/loss"])) <= 1e-5 * abs(float(g[f"{tag}/loss"])) + 1e-6
    sl = slice(None) if k <= 7 else slice(0, 32)
    assert rel_err(q.grad[:, sl], g[f"{tag}/dq"]) < 1e-4
    assert rel_err(p.grad[:, sl], g[f"{tag}/dp"]) < 1e-4


def test_infonce_validation():
    with pytest.raises(ValueError):
        R.info_nce(torch.zeros(2, 3, 4), torch.zeros(2, 4))
    with pytest.raises(ValueError):
        R.info_nce(torch.zeros(2, 4), torch.zeros(3, 4))
    with pytest.raises(ValueError):
        R.info_nce(torch.zeros(2, 4), torch.zeros(2, 5))


@pytest.mark.parametrize("k", [2, 7, 32])
def test_got(k):
    g = golden("got")
    N = 40
    trial = int(g[f"k{k}/trial"])  # first well-conditioned recipe key, see oracle/gen_golden.py
    v0, q0 = t((k, N, 128), f"got:v{k}:{trial}"), t((k, N, 128), f"got:q{k}:{trial}")
    q0 = q0 + 0.7 * v0
    v, q = v0.clone().requires_grad_(), q0.clone().requires_grad_()
    loss = R.got(v, q, subsample=256, perm=torch.from_numpy(g[f"k{k}/perm"]))
    loss.backward()
    assert abs(float(loss) - float(g[f"k{k}/loss"])) < 1e-5 * abs(float(g[f"k{k}/loss"]))
    assert abs(float(v.grad.norm()) - float(g[f"k{k}/dv_norm"])) < 1e-4 * float(g[f"k{k}/dv_norm"])
    assert abs(float(q.grad.norm()) - float(g[f"k{k}/dq_norm"])) < 1e-4 * float(g[f"k{k}/dq_norm"])
    assert float(v.grad[:, k:].abs().max()) == 0.0 == float(g[f"k{k}/dv_tail_abs"])  # randperm(batch) quirk
    if k <= 7:
        assert rel_err(v.grad[:, :k], g[f"k{k}/dv"]) < 1e-4
        assert rel_err(q.grad[:, :k], g[f"k{k}/dq"]) < 1e-4
    else:
        assert rel_err(v.grad[:4, :k, :16], g[f"k{k}/dv"]) < 1e-4
        assert rel_err(q.grad[:4, :k, :16], g[f"k{k}/dq"]) < 1e-4
    # the permutation only changes summation order: identity perm gives the same value
    loss_id = R.got(v0, q0, subsample=256, perm=torch.arange(k))
    assert abs(float(loss_id) - float(g[f"k{k}/loss"])) < 1e-5 * abs(float(g[f"k{k}/loss"]))


def test_got_pieces():
    g = golden("got")
    k, n = 3, 9
    trial = int(g["piece/trial"])
    v = t((k, n, 128), f"got:pv:{trial}")
    q = t((k, n, 128), f"got:pq:{trial}") + 0.5 * v
    c = R.cross_cost(v, q)
    assert max_rel(c, g["piece/cross_cost"]) < 1e-5
    assert max_rel(R.intra_cost(v), g["piece/intra_cost"], floor=1e-3) < 1e-4
    cthr = R.threshold_relu(c)
    tp = R.ipot(cthr, 0.5, 30)
    assert max_rel(tp, g["piece/ipot30"]) < 1e-4
    assert rel_err((cthr * tp).sum(dim=(1, 2)), g["piece/wd"][:, 0]) < 1e-5
    assert rel_err(R.gw_distance(v, q), g["piece/gwd"][:, 0]) < 1e-4
    assert abs(float(R.got(v, q)) - float(g["nosub/loss"])) < 1e-5 * abs(float(g["nosub/loss"]))


def _cl_inputs():
    B, M, N = 6, 5, 12
    stains = MODS5[1:]
    he_e, he_t = t((B, 1, 512), "cl:he_e"), t((B, N, 128), "cl:he_t")
    wsi = {"HE": he_e.unsqueeze(3).repeat(1, 1, 1, M - 1)}
    tok = {"HE": he_t.unsqueeze(3).repeat(1, 1, 1, M - 1)}
    for s in stains:
        wsi[s] = t((B, 1, 512), f"cl:e{s}") + 0.1 * he_e
        tok[s] = t((B, N, 128), f"cl:t{s}") + 0.6 * he_t
    return B, M, N, stains, wsi, tok


def test_calculate_losses():
    g = golden("calculate_losses")
    B, M, N, stains, wsi, tok = _cl_inputs()
    labels = torch.from_numpy(g["labels"])
    nce = lambda a, b, symmetric=False: R.info_nce(a, b, 0.001, symmetric)  # noqa: E731
    wl = {k: v.clone().requires_grad_() for k, v in wsi.items()}
    tl = {k: v.clone().requires_grad_() for k, v in tok.items()}
    torch.manual_seed(5)
    loss, flag = R.calculate_losses(stains, nce, R.got, None, wl, tl, labels[:, 1:], True, 0.7)
    loss.backward()
    assert flag and bool(g["full/flag"])
    assert abs(float(loss) - float(g["full/loss"])) < 1e-5 * abs(float(g["full/loss"]))
    for k in wsi:
        gn = 0.0 if wl[k].grad is None else float(wl[k].grad.norm())
        assert abs(gn - float(g[f"full/dwsi_norm/{k}"])) <= 1e-4 * float(g[f"full/dwsi_norm/{k}"]) + 1e-9
        gn = 0.0 if tl[k].grad is None else float(tl[k].grad.norm())
        assert abs(gn - float(g[f"full/dtok_norm/{k}"])) <= 1e-4 * float(g[f"full/dtok_norm/{k}"]) + 1e-9
    loss_g, _ = R.calculate_losses(stains, nce, None, None, wsi, tok, labels[:, 1:], True, 0.7)
    assert abs(float(loss_g) - float(g["global/loss"])) < 1e-5 * abs(float(g["global/loss"]))
    l0 = torch.zeros(B, M)
    l0[:, 0] = 1
    l0[2, 3] = 1
    loss_s, flag_s = R.calculate_losses(stains, nce, R.got, None, wsi, tok, l0[:, 1:], True, 0.7)
    assert loss_s == -1 and flag_s is False and float(g["sentinel/loss"]) == -1.0
    wsi3 = {k: torch.cat([v, t(v.shape, f"cl:v1{k}"), t(v.shape, f"cl:v2{k}")], dim=1) for k, v in wsi.items()}
    loss_i, _ = R.calculate_losses(stains, nce, None, nce, wsi3, tok, labels[:, 1:], True, 0.7)
    assert abs(float(loss_i) - float(g["intra/loss"])) < 1e-5 * abs(float(g["intra/loss"]))


def test_full_step():
    g = golden("full_step")
    B, M, N, D = (int(x) for x in g["shape"])
    mods = MODS5[:M]
    labels = torch.from_numpy(g["labels"])
    feats = t((B, M, N, D), "fs:feats")
    sd = recipe_params(M, D, "wfs", requires_grad=True)
    torch.manual_seed(11)
    loss, flag, _ = R.pretrain_step_loss(feats, labels, sd, mods, 0.001, True, use_got=True)
    loss.backward()
    assert flag
    assert abs(float(loss) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    _grads_match(g, sd, tol=2e-4)
    sd2 = recipe_params(M, D, "wfs", requires_grad=True)
    loss_g, _, _ = R.pretrain_step_loss(feats, labels, sd2, mods, 0.001, True, use_got=False)
    loss_g.backward()
    assert abs(float(loss_g) - float(g["global/loss"])) < 1e-5 * abs(float(g["global/loss"]))
    _grads_match(g, sd2, prefix="global/", tol=2e-4)
