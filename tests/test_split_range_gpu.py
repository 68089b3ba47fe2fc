"""Accuracy contract of the split-fp16 engine on ADVERSARIAL dynamic range.

The split engine carries an fp32 tensor as two fp16 planes under a power-of-two scale; a value 2^-k below the scale's maximum keeps
~38 - k bits.  True fp32 (the reference: nn.Linear / F.layer_norm / autograd, madeleine/models/Model.py:350-363, abmil.py:49-52) has no
such coupling between the elements of a tensor.  These tests put the coupling under stress -- one patch / one channel 2^20 and 2^24
times the bulk, Student-t(2) features, all-zero bags mixed with live ones (wsi_dataset.py:66), rows spread over 30 binades, a softmax
with one dominant token per bag -- and compare BOTH GEMM modes (split; the exact-fp32 matrix-core kernels, MADELEINE_GEMM=fp32) with
an fp64 evaluation, norm-relative AND per-row max-relative.  Criterion everywhere: the split mode's error is within 2x the exact-fp32
kernels' error (plus a floor of a few fp32 ulps), i.e. on these inputs it is as good an fp32 implementation as the fp32 kernels are.
tools/split_range_report.py dumps the measured table (profiles/r04_split_range_report.json).
"""
from types import SimpleNamespace

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import restatement as R
from tests._util import MODS5, t
from tests.test_hip_kernels import _gate_weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


# ------------------------------------------------------------------------------------------------- metrics
def norm_rel(a, ref):
    a, ref = a.detach().double().cpu(), ref.detach().double().cpu()
    return float((a - ref).norm() / ref.norm().clamp_min(1e-300))


def row_rel(a, ref):
    """max over rows of  max_j |a - ref| / max_j |ref|  (rows whose reference is identically zero must be zero)."""
    a, ref = a.detach().double().cpu(), ref.detach().double().cpu()
    a, ref = a.reshape(-1, a.shape[-1]), ref.reshape(-1, ref.shape[-1])
    top = ref.abs().amax(1)
    err = (a - ref).abs().amax(1)
    live = top > 0
    assert float(err[~live].max() if (~live).any() else 0.0) == 0.0
    return float((err[live] / top[live]).max())


def row_envelope(a_split, a_f32, ref, floor=4e-6):
    """Per-row contract of a GRADIENT tensor produced from a split image with one scale per tensor: every row is either as good as the
    exact-fp32 kernels' (4x their error -- two fp32-class results, row maxima of rounding noise --, floor = a few fp32 ulps of the row's
    largest entry), or its absolute error is below 2^-26 of
    the tensor's largest entry (image: 2^-38 of the scale bound per element; the bound's slack over the true maximum <= 2^6; a 1024-
    long contraction with the gate weights; measured worst 2^-27.6) -- rows 2^-k below the largest row keep ~36 - k bits instead of
    fp32's 24, which is invisible in every sum over rows (dW, dgamma, dbeta, embedding gradients: all that such rows enter).
    Returns max_r err_r / allowed_r (<= 1)."""
    a_split, a_f32, ref = (v.detach().double().cpu().reshape(-1, v.shape[-1]) for v in (a_split, a_f32, ref))
    top = ref.abs().amax(1)
    e_s, e_f = (a_split - ref).abs().amax(1), (a_f32 - ref).abs().amax(1)
    allowed = 4.0 * torch.maximum(e_f, floor * top) + 2.0 ** -26 * float(top.max())
    return float((e_s / allowed.clamp_min(1e-300)).max())


def within(e_split, e_f32, floor):
    return e_split <= max(2.0 * e_f32, floor)


# ------------------------------------------------------------------------------------------------- inputs
KINDS = ["uniform", "outlier_patch_2^20", "outlier_patch_2^24", "outlier_channel_2^20", "student_t2", "zero_bags", "rows_over_30_binades"]


def make_x(kind, T, K):
    x = t((T, K), f"rng:x:{kind}") * 2
    if kind.startswith("outlier_patch"):
        x[T // 3] *= 2.0 ** int(kind.split("^")[1])
    elif kind.startswith("outlier_channel"):
        x[:, 17] *= 2.0 ** 20
    elif kind == "student_t2":
        g = torch.Generator().manual_seed(5)
        z = torch.randn(T, K, generator=g, dtype=torch.float64)
        chi = torch.randn(T, K, 2, generator=g, dtype=torch.float64).square().sum(-1) / 2.0
        x = (z / chi.sqrt()).float()                      # infinite variance: a handful of entries in the thousands
    elif kind == "zero_bags":
        x[T // 4: 3 * T // 4] = 0.0                       # absent-stain bags (wsi_dataset.py:66) between live ones
    elif kind == "rows_over_30_binades":
        x = x * torch.logspace(0, -9, T).unsqueeze(1)
    return x.contiguous()


# ------------------------------------------------------------------------------------------------- case 1: first pre_attn block
def case_block1(dev, kind, T=1200, K=512, N=512, with_bias=True):
    """Linear(K -> N) -> LayerNorm -> GELU on the CALLER's rows (the patch features) in both GEMM modes against fp64: output, dX, dW,
    dbias, dgamma, dbeta."""
    from madeleine_amd import functional as MF
    x = make_x(kind, T, K)
    W = 0.05 * t((N, K), "rng:w")
    lb = 0.3 * t((N,), "rng:lb") if with_bias else None
    g, b = 1 + 0.2 * t((N,), "rng:g"), 0.3 * t((N,), "rng:b")
    dy = t((T, N), "rng:dy") * torch.logspace(0, -2, T).unsqueeze(1)
    names = ["x", "W", "lin_bias", "gamma", "beta"] if with_bias else ["x", "W", "gamma", "beta"]
    src = [x, W, lb, g, b] if with_bias else [x, W, g, b]
    lv = [v.double().requires_grad_() for v in src]
    pre = lv[0] @ lv[1].t() + (lv[2] if with_bias else 0.0)
    ref = F.gelu(F.layer_norm(pre, (N,), lv[-2], lv[-1], 1e-5))
    ref.backward(dy.double())
    res = {}
    old = MF.gemm_mode()
    try:
        for mode in ("split", "fp32"):
            MF.set_gemm_mode(mode)
            dl = [v.to(dev).requires_grad_() for v in src]
            xd, Wd = dl[0], dl[1]
            lbd = dl[2] if with_bias else None
            gd, bd = dl[-2], dl[-1]
            if mode == "split":
                _img, _sc, out = MF.preattn_block(xd, None, Wd, lbd, gd, bd, 1e-5, 0.0, 0, None, True)
            else:
                out = MF.ln_gelu_drop(MF.linear(xd, Wd), gd, bd, 1e-5, 0.0, 0, None, lbd)
            out.backward(dy.to(dev))
            r = {"out_norm": norm_rel(out, ref), "out_row": row_rel(out, ref), "dx_row": row_rel(dl[0].grad, lv[0].grad)}
            for n_, a_, r_ in zip(names, dl, lv):
                r["d" + n_] = norm_rel(a_.grad, r_.grad)
            assert all(np.isfinite(v) for v in r.values()), (mode, r)
            res[mode] = r
            res[mode + "_dx"] = dl[0].grad.detach().cpu()
    finally:
        MF.set_gemm_mode(old)
    res["dx_envelope"] = row_envelope(res.pop("split_dx"), res.pop("fp32_dx"), lv[0].grad)
    return res


@pytest.mark.parametrize("kind", KINDS)
def test_first_block_on_adversarial_rows(dev, kind):
    res = case_block1(dev, kind)
    s, f = res["split"], res["fp32"]
    for key in s:
        if key == "dx_row":     # per-row gradient quantity: the envelope below (recorded in the report)
            continue
        # floors: a few fp32 ulps on the forward values (PER ROW: the row-scaled image of the caller's features); 2e-6 on gradient norms
        floor = 1e-6 if key.startswith("out") else 2e-6
        assert within(s[key], f[key], floor), (kind, key, s[key], f[key])
    assert res["dx_envelope"] <= 1.0, (kind, res["dx_envelope"], s["dx_row"], f["dx_row"])


def test_first_block_zero_rows_without_bias(dev):
    """All-zero bags with a bias-free Linear: constant pre-LN rows, rstd = 1 / sqrt(eps) = 316 on those rows only -- the dx-image bound
    (max_r rstd[r] row_mul[r]) must not cost the live rows their bits."""
    res = case_block1(dev, "zero_bags", with_bias=False)
    s, f = res["split"], res["fp32"]
    for key in s:
        if key != "dx_row":
            assert within(s[key], f[key], 4e-6), (key, s[key], f[key])
    assert res["dx_envelope"] <= 1.0, (res["dx_envelope"], s["dx_row"], f["dx_row"])


# ------------------------------------------------------------------------------------------------- case 2: peaked attention
def case_attnpool(dev, peak, BM=3, N=700, H=4):
    """Fused A2 + A3 (gate scores -> softmax over patches -> pooling) forward + backward with wc scaled so that the softmax has one
    dominant token per bag (what a trained head produces): d_scores then spans > 20 binades across the tokens of a bag."""
    from madeleine_amd import functional as MF
    E = t((BM, N, H * 512), "rng:E") * 1.5
    wa, ba, wb, bb, wc, bc = _gate_weights(H, "rng:gw")
    wc = wc * peak
    dp = t((BM, H * 512), "rng:dp")
    src = [E, wa, ba, wb, bb, wc, bc]
    names = ["E", "Wa", "ba", "Wb", "bb", "wc", "bc"]
    lv = [v.double().requires_grad_() for v in src]
    x = lv[0].view(BM, N, H, 512)
    a = torch.tanh(torch.einsum("bnhe,hfe->bnhf", x, lv[1]) + lv[2])
    b = torch.sigmoid(torch.einsum("bnhe,hfe->bnhf", x, lv[3]) + lv[4])
    sc = ((a * b) * lv[5]).sum(-1) + lv[6]                                  # [BM,N,H]
    w = torch.softmax(sc, dim=1)
    ref = torch.einsum("bnh,bnhe->bhe", w, x).reshape(BM, H * 512)
    ref.backward(dp.double())
    res = {"w_max": float(w.detach().amax(1).mean()), "dscore_binades": None}
    old = MF.gemm_mode()
    try:
        for mode in ("split", "fp32"):
            MF.set_gemm_mode(mode)
            dl = [v.to(dev).requires_grad_() for v in src]
            pooled, scores = MF.attn_pool(*dl)
            pooled.backward(dp.to(dev))
            r = {"pooled": norm_rel(pooled, ref), "scores": norm_rel(scores.view(BM, N, H), sc),
                 "dE_norm": norm_rel(dl[0].grad, lv[0].grad), "dE_row": row_rel(dl[0].grad, lv[0].grad)}
            for n_, a_, r_ in zip(names[1:], dl[1:], lv[1:]):
                r["d" + n_] = norm_rel(a_.grad, r_.grad)
            assert all(np.isfinite(v) for v in r.values()), (mode, r)
            res[mode] = r
            res[mode + "_dE"] = dl[0].grad.detach().cpu()
    finally:
        MF.set_gemm_mode(old)
    res["dE_envelope"] = row_envelope(res.pop("split_dE"), res.pop("fp32_dE"), lv[0].grad)
    return res


@pytest.mark.parametrize("peak", [1.0, 40.0, 120.0])
def test_attnpool_backward_with_dominant_tokens(dev, peak):
    res = case_attnpool(dev, peak)
    s, f = res["split"], res["fp32"]
    if peak >= 40:
        assert res["w_max"] > 0.3          # the softmax really is peaked
    for key in s:
        if key == "dbc":                   # shift invariance of the softmax: the exact gradient is 0, both sides return rounding noise
            continue
        if key == "dE_row":
            # per-row: rows whose gradient is 2^-k of the largest row's keep ~38 - k bits in the dz image (one scale per tensor); the
            # exact-fp32 kernels keep 24 -- row_envelope; such rows enter nothing but sums over tokens (dW, dgamma, ...)
            continue
        assert within(s[key], f[key], 3e-6), (peak, key, s[key], f[key])
    if f["dE_row"] < 0.1:    # (beyond that the softmax weights of the non-dominant tokens underflow fp32 itself: rows lost on both sides)
        assert res["dE_envelope"] <= 1.0, (peak, res["dE_envelope"], s["dE_row"], f["dE_row"])


# ------------------------------------------------------------------------------------------------- case 3: the whole step at T = 0.001
def case_full_step(dev, kind, use_got=False, T_=0.001):
    """Encoder + InfoNCE (T = 0.001, the training value) (+ GOT) + backward on adversarial bags, both modes against the fp64 oracle."""
    from madeleine_amd import GOT, InfoNCE, calculate_losses
    from madeleine_amd import functional as MF
    from tests.test_model_gpu import build
    B, M, N, D = 8, 3, 160, 64
    mods = MODS5[:M]
    feats = make_x(kind, B * M * N, D).view(B, M, N, D) * 0.5
    labels = torch.ones(B, M)
    if kind == "zero_bags":
        feats = t((B, M, N, D), "rng:fs") * 1.0
        labels[1, 1] = labels[5, 2] = labels[6, 1] = 0
        feats = feats * labels[:, :, None, None]
    args = SimpleNamespace(global_loss="info-nce", symmetric_cl=True, local_loss_weight=0.5)
    model = build(mods, D, "wrng", dev).eval()
    sd64 = {k: v.detach().cpu().double().requires_grad_() for k, v in model.state_dict().items()}
    torch.manual_seed(3)
    ref_loss, flag, _ = R.pretrain_step_loss(feats.double(), labels, sd64, mods, T_, True, use_got=use_got, local_weight=0.5)
    ref_loss.backward()
    res = {}
    old = MF.gemm_mode()
    try:
        for mode in ("split", "fp32"):
            MF.set_gemm_mode(mode)
            model.zero_grad()
            embs, toks = model({"feats": feats}, device=dev, train=True)
            torch.manual_seed(3)
            loss, flag = calculate_losses(mods[1:], InfoNCE(temperature=T_), GOT if use_got else None, None, embs, toks, labels[:, 1:],
                                          args)
            loss.backward()
            top = max(float(v.grad.norm()) for v in sd64.values() if v.grad is not None)
            r = {"loss": abs(float(loss) - float(ref_loss)) / abs(float(ref_loss))}
            worst = 0.0
            for k, p in model.named_parameters():
                ref = sd64[k].grad
                if ref is None or p.grad is None:
                    continue
                worst = max(worst, float((p.grad.detach().double().cpu() - ref).norm()) / max(float(ref.norm()), 1e-3 * top))
            r["worst_param_grad"] = worst
            assert np.isfinite(r["loss"]) and np.isfinite(worst), (mode, r)
            res[mode] = r
    finally:
        MF.set_gemm_mode(old)
    return res


@pytest.mark.parametrize("kind", ["uniform", "outlier_patch_2^20", "student_t2", "zero_bags"])
def test_full_step_at_training_temperature(dev, kind):
    """At T = 0.001 the InfoNCE softmax is saturated for small batches: softmax - onehot cancels and the fp32 gradient is only as
    good as its forward values (tests/test_hip_kernels.py:_grad_ok) -- which is exactly where an operand format that lost bits would
    show.  Criterion: within 1e-3 of fp64 (north_star), or within 2x of what the exact-fp32 kernels achieve."""
    res = case_full_step(dev, kind)
    s, f = res["split"], res["fp32"]
    assert s["loss"] <= max(1e-3, 2 * f["loss"]), (kind, s, f)
    assert s["worst_param_grad"] <= max(1e-3, 2 * f["worst_param_grad"]), (kind, s, f)


def test_full_step_with_got_on_outlier_bags(dev):
    res = case_full_step(dev, "outlier_patch_2^20", use_got=True, T_=0.01)
    s, f = res["split"], res["fp32"]
    assert s["loss"] <= max(1e-3, 2 * f["loss"]) and s["worst_param_grad"] <= max(1e-3, 2 * f["worst_param_grad"]), (s, f)


# ------------------------------------------------------------------------------------------------- published bounds
def test_second_consumer_of_E_does_not_use_a_stale_bound(dev):
    """max|dE| travels from the gate backward's epilogue to the LayerNorm backward as a scale bound.  With a
    SECOND autograd consumer of E whose gradient is 10^4 times larger (return_preattn_feats=True tokens in a loss) the autograd
    engine sums the two gradients before the LayerNorm backward runs: the published maximum no longer bounds the tensor.  The entry
    is bound to the buffer's identity and version (functional._take_absmax): the consumer must notice and take its own absmax --
    a stale bound here would overflow the fp16 hi plane (inf / nan gradients)."""
    from madeleine_amd import functional as MF
    from madeleine_amd.model import ABMILEmbedder
    torch.manual_seed(3)
    emb = ABMILEmbedder(pre_attention_params={"input_dim": 512, "hidden_dim": 512}, attention_params={
        "model": "ABMIL", "params": {"input_dim": 512, "hidden_dim": 512, "dropout": False, "activation": "softmax", "n_heads": 4,
                                     "n_classes": 1}}).to(dev).eval()
    bags = t((2, 400, 512), "rng:bags2").to(dev)
    res = {}
    old = MF.gemm_mode()
    try:
        for mode in ("fp32", "split"):
            MF.set_gemm_mode(mode)
            emb.zero_grad()
            slide, tokens = emb(bags, return_preattn_feats=True)
            (slide.sum() + 1e4 * tokens.square().sum()).backward()
            res[mode] = {k: p.grad.detach().clone() for k, p in emb.named_parameters() if p.grad is not None}
    finally:
        MF.set_gemm_mode(old)
    for k, g in res["fp32"].items():
        assert torch.isfinite(res["split"][k]).all(), k
        if g.numel() == 1:      # attention_c.bias: shift invariance of the softmax, the exact gradient is 0
            continue
        # (both modes round the 10^4-times larger token term into dE: the slide-path gradients agree to ~1e-4 of themselves)
        assert norm_rel(res["split"][k], g) < 2e-3, (k, norm_rel(res["split"][k], g))


def test_split_nodes_survive_retain_graph(dev):
    """the split images travel through ctx.save_for_backward, so a second backward over a retained graph runs
    the same kernels and returns the same bits (it used to crash in SplitLinearFn / switch engines in the gate nodes)."""
    from madeleine_amd import functional as MF
    if MF.gemm_mode() != "split":
        pytest.skip("split GEMM mode only")
    x = t((600, 512), "rng:rg:x").to(dev).requires_grad_()
    W = (0.05 * t((512, 512), "rng:rg:w")).to(dev).requires_grad_()
    y = MF.linear(x, W)
    g = []
    for _ in range(2):
        x.grad = W.grad = None
        y.sum().backward(retain_graph=True)
        g.append((x.grad.clone(), W.grad.clone()))
    assert torch.equal(g[0][0], g[1][0]) and torch.equal(g[0][1], g[1][1])
    E = (t((2, 300, 2048), "rng:rg:E") * 1.5).to(dev).requires_grad_()
    w = [v.to(dev).requires_grad_() for v in _gate_weights(4, "rng:rg:gw")]
    pooled, _ = MF.attn_pool(E, *w)
    g = []
    for _ in range(2):
        E.grad = None
        pooled.square().sum().backward(retain_graph=True)
        g.append(E.grad.clone())
    assert torch.equal(g[0], g[1])


def test_engine_precision_by_binades_below_the_scale(dev):
    """The engine's documented contract, measured through the matrix cores (fp16 subnormals of the lo plane included): a row 2^-k below
    the image's scale comes out of a product with relative error <= max(2^-20, 2^(k - 37)) -- full fp32 precision down to 2^-17 of the
    tensor maximum, one bit lost per binade below that (include/madeleine_amd.h: relative representation error <= max(2^-23,
    2^-25 / |scaled value|))."""
    from madeleine_amd import functional as MF
    M, N, K = 40, 256, 512
    a = t((M, K), "bin:a") + 1.5 * torch.sign(t((M, K), "bin:a"))     # magnitudes in [0.5, 2.5]
    a = a * (2.0 ** -torch.arange(M, dtype=torch.float32)).unsqueeze(1)
    b = 0.05 * t((N, K), "bin:b")
    ref = a.double() @ b.double().t()
    C = MF.split_gemm_nt(MF.split_image(a.to(dev)), MF.split_image(b.to(dev))).double().cpu()
    for k in range(M):
        e = float((C[k] - ref[k]).abs().max() / ref[k].abs().max())
        assert e <= max(2.0 ** -20, 2.0 ** (k - 37)), (k, e)
    # the row-scaled image has no such coupling: every row at full precision
    Ai, row_inv = MF.split_image_rows(a.to(dev))
    C2 = MF.split_gemm_nt(Ai, MF.split_image(b.to(dev)), a_row_mul=row_inv).double().cpu()
    for k in range(M):
        assert float((C2[k] - ref[k]).abs().max() / ref[k].abs().max()) <= 2.0 ** -20, k


def test_gate_activation_recompute_is_bit_identical(dev):
    """Opt-in recomputation of the saved gate activations (functional.set_gate_recompute): nothing but E, the parameters and the
    softmax statistics is kept by the fused A2 + A3 node; the backward re-runs the gate forward with the forward's seed.  Same bits
    as the default path, dropout on, both GEMM modes."""
    from madeleine_amd import functional as MF
    E0 = (t((3, 500, 2048), "rng:rc:E") * 1.5)
    w0 = _gate_weights(4, "rng:rc:gw")
    dp = t((3, 2048), "rng:rc:dp").to(dev)
    old = MF.gemm_mode()
    try:
        for mode in ("split", "fp32"):
            MF.set_gemm_mode(mode)
            res = []
            for rc in (False, True):
                MF.set_gate_recompute(rc)
                E = E0.to(dev).requires_grad_()
                w = [v.to(dev).requires_grad_() for v in w0]
                torch.cuda.reset_peak_memory_stats()
                base = torch.cuda.memory_allocated()
                pooled, scores = MF.attn_pool(E, *w, p_drop=0.25, seed=1234)
                held = torch.cuda.memory_allocated() - base
                pooled.backward(dp)
                res.append((pooled.detach().clone(), E.grad.clone(), [v.grad.clone() for v in w], held))
            assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]), mode
            for a, b in zip(res[0][2], res[1][2]):
                assert torch.equal(a, b), mode
            assert res[1][3] < res[0][3] - 2 * 1500 * 4 * 512 * 4 * 0.9, (mode, res[0][3], res[1][3])   # the two activation tensors are gone
    finally:
        MF.set_gate_recompute(False)
        MF.set_gemm_mode(old)


def test_weight_image_keeps_every_output_channel_at_full_precision(dev):
    """functional.weight_image: one power-of-two scale per ROW of W (= per output channel of nn.Linear, Model.py:351), applied as a
    per-column factor in the NT epilogue -- output channels whose weights are 2^-30 of the largest row's keep full precision
    (a per-tensor scale would leave them ~8 bits), an all-zero weight row gives exactly bias, and the product equals the fp64 one."""
    from madeleine_amd import functional as MF
    M, N, K = 700, 256, 512
    x = t((M, K), "wimg:x") * 2
    W = 0.05 * t((N, K), "wimg:w") * (2.0 ** -torch.linspace(0, 30, N)).unsqueeze(1)
    W[17] = 0.0
    bias = 0.3 * t((N,), "wimg:b")
    Wi = MF.weight_image(W.to(dev))
    assert Wi.row_inv is not None and float(Wi.row_inv[17]) == 0.0 and float(Wi.scale[0]) == 1.0
    xi, row_inv = MF.split_image_rows(x.to(dev))
    Cb = MF.split_gemm_nt(xi, Wi, bias.to(dev), a_row_mul=row_inv).double().cpu()
    assert float((Cb[:, 17] - bias[17].double()).abs().max()) == 0.0
    C = MF.split_gemm_nt(xi, Wi, a_row_mul=row_inv).double().cpu()
    lin = x.double() @ W.double().t()
    assert float((Cb - (lin + bias.double())).abs().max()) < 1e-6 * float(lin.abs().max())
    col_err = ((C - lin).abs().amax(0) / lin.abs().amax(0).clamp_min(1e-300))
    col_err[17] = 0.0
    assert float(col_err.max()) < 2.0 ** -20, float(col_err.max())
    with pytest.raises(ValueError):
        MF.split_gemm_tn(xi, Wi)      # a row-scaled image cannot be contracted over its rows
