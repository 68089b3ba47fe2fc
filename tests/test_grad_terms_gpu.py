"""Two-term backward products of the split engine (functional.set_gradient_terms(2); include/madeleine_amd.h `terms`): the operand
that is not a gradient -- the weight of dX = dY W, the activations of dW = dY^T X (reference: the autograd of nn.Linear,
madeleine/models/Model.py:351-359, and of the gated attention, Model.py:27-34) -- enters with its hi plane only.  Pinned here:
what exactly is computed (the 3-term product of the hi-plane-rounded operand), that it is bit-identical to 3 terms when that operand is
fp16-exact, that the forward is untouched, and how far the gradients of a full step move."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from tests._util import rel_err, t

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _hi_plane(img, rows, K):
    """fp64 value of the hi plane alone: RN16(scale x) / scale (times the row factor of a row-scaled image)."""
    raw = img.data[:rows].contiguous().view(torch.int16).view(rows, K // 32, 2, 32).view(torch.float16).double()
    v = raw[:, :, 0].reshape(rows, K) / float(img.scale[0])
    if img.row_inv is not None:
        v = v * img.row_inv.double().unsqueeze(1)
    return v.cpu()


@pytest.mark.parametrize("M,N,K", [(700, 512, 512), (513, 800, 2048), (257, 4, 32), (3000, 128, 2048)])
def test_nt_two_terms_is_the_product_with_the_hi_plane_of_b(dev, M, N, K):
    from madeleine_amd import functional as MF
    a = t((M, K), f"gt:a{M}{K}") * torch.logspace(0, -3, M).unsqueeze(1)           # a gradient-like A: rows 1000x apart
    b = 0.05 * t((N, K), f"gt:b{N}{K}")
    A, B = MF.split_image(a.to(dev)), MF.weight_image(b.to(dev))
    C2 = MF.split_gemm_nt(A, B, terms=2)
    C3 = MF.split_gemm_nt(A, B, terms=3)
    exact = a.double() @ b.double().t()
    assert rel_err(C3, exact) < 1e-6
    assert rel_err(C2, a.double() @ _hi_plane(B, N, K).t()) < 1e-6                # A keeps both planes, B is its hi plane
    e2 = rel_err(C2, exact)
    assert 1e-6 < e2 < 2.0 ** -11, e2                                              # 11-bit operand: ~2^-12.5 rms
    # fp16-exact B (lo plane identically zero): dropping ah bl drops exact zeros -> the same bits
    bq = (b * 64).half().float() / 64
    Bq = MF.split_image(bq.to(dev))
    assert torch.equal(MF.split_gemm_nt(A, Bq, terms=2), MF.split_gemm_nt(A, Bq, terms=3))


@pytest.mark.parametrize("T,Mi,N", [(3000, 512, 512), (12325, 512, 1024), (33, 32, 32)])
def test_tn_two_terms_is_the_product_with_the_hi_plane_of_a(dev, T, Mi, N):
    from madeleine_amd import functional as MF
    x = t((T, Mi), f"gt:x{T}{Mi}") * 2
    dy = t((T, N), f"gt:d{T}{N}") * torch.logspace(0, -3, T).unsqueeze(1)
    X, D = MF.split_image(x.to(dev)), MF.split_image(dy.to(dev), pad_rows=32)
    W2 = MF.split_gemm_tn(X, D, terms=2)
    W3 = MF.split_gemm_tn(X, D, terms=3)
    exact = dy.double().t() @ x.double()
    assert rel_err(W3, exact) < 1e-6
    assert rel_err(W2, dy.double().t() @ _hi_plane(X, T, Mi)) < 1e-6
    e2 = rel_err(W2, exact)
    assert 1e-6 < e2 < 2.0 ** -11, e2
    xq = (x * 16).half().float() / 16
    Xq = MF.split_image(xq.to(dev))
    assert torch.equal(MF.split_gemm_tn(Xq, D, terms=2), MF.split_gemm_tn(Xq, D, terms=3))


def test_terms_argument_is_checked(dev):
    from madeleine_amd import functional as MF
    A = MF.split_image(t((64, 32), "gt:ea").to(dev))
    with pytest.raises(RuntimeError):
        MF.split_gemm_nt(A, A, terms=1)
    with pytest.raises(ValueError):
        MF.set_gradient_terms(4)


def _step(dev, terms, use_got):
    from madeleine_amd import GOT, InfoNCE, calculate_losses
    from madeleine_amd import functional as MF
    from tests.test_model_gpu import build
    MODS = ["HE", "ER", "PR", "KI67", "HER2"]
    B, M, N, D = 8, 3, 300, 64
    feats = t((B, M, N, D), "gt:fs") * 0.5
    labels = torch.ones(B, M)
    args = SimpleNamespace(global_loss="info-nce", symmetric_cl=True, local_loss_weight=0.5)
    model = build(MODS[:M], D, "wgt", dev).train()
    MF.set_gradient_terms(terms)
    try:
        torch.manual_seed(11)
        embs, toks = model({"feats": feats}, device=dev, train=True)
        torch.manual_seed(3)
        loss, _ = calculate_losses(MODS[1:M], InfoNCE(temperature=0.001), GOT if use_got else None, None, embs, toks, labels[:, 1:], args)
        loss.backward()
    finally:
        MF.set_gradient_terms(3)
    return loss.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}


@pytest.mark.parametrize("use_got", [False, True])
def test_full_step_forward_untouched_gradients_at_eleven_bits(dev, use_got):
    """Dropout on, T = 0.001: the loss is the same bits with 2 and 3 gradient terms (only backward products change); every parameter
    gradient stays within 1e-3 (norm-relative) of the 3-term gradient -- the tolerance BASELINE.json states for the path."""
    l3, g3 = _step(dev, 3, use_got)
    l2, g2 = _step(dev, 2, use_got)
    assert torch.equal(l2, l3)
    top = max(float(v.norm()) for v in g3.values())
    worst, moved = 0.0, False
    for k, v in g3.items():
        d = float((g2[k] - v).norm()) / max(float(v.norm()), 1e-3 * top)
        worst = max(worst, d)
        moved = moved or d > 0
    assert moved and worst < 1e-3, worst
    assert np.isfinite(worst)


def test_calculate_losses_batches_the_local_terms_without_changing_them(dev):
    """trainer.calculate_losses (reference trainer.py:20-77) runs the stains' GOT terms as one node of concurrent chains
    (distributed.got_multi(local=True)); a callable that is not madeleine_amd.GOT itself takes the per-stain route.  Same randperm
    draws, same thresholds: loss and every parameter gradient agree to fp32 rounding."""
    from madeleine_amd import GOT, InfoNCE, calculate_losses
    from tests.test_model_gpu import build
    MODS = ["HE", "ER", "PR", "KI67", "HER2"]
    B, M, N, D = 10, 4, 200, 64
    feats = t((B, M, N, D), "bl:fs") * 0.5
    labels = torch.ones(B, M)
    labels[2, 1] = labels[7, 3] = labels[8, 3] = 0
    feats = feats * labels[:, :, None, None]
    args = SimpleNamespace(global_loss="info-nce", symmetric_cl=True, local_loss_weight=0.7)
    model = build(MODS[:M], D, "wbl", dev).eval()
    res = []
    for local in (GOT, lambda v, q, subsample=None: GOT(v, q, subsample=subsample)):
        model.zero_grad()
        embs, toks = model({"feats": feats}, device=dev, train=True)
        torch.manual_seed(5)
        loss, flag = calculate_losses(MODS[1:M], InfoNCE(temperature=0.1), local, None, embs, toks, labels[:, 1:], args)
        assert flag
        loss.backward()
        res.append((loss.detach().clone(), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}))
    assert abs(float(res[0][0]) - float(res[1][0])) <= 2e-6 * abs(float(res[1][0]))
    top = max(float(v.norm()) for v in res[1][1].values())
    for k, v in res[1][1].items():
        assert float((res[0][1][k] - v).norm()) <= 2e-5 * max(float(v.norm()), 1e-3 * top), k
